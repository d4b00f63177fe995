#!/usr/bin/env python3
"""Phase breakdown of k_match_pass (passes 0 + 1 in one launch) from the stamps of the
measurement build (`make probe` -> yadcc_amd/libydc_probe.so; YDC_PHASE_PROBE in match_kernel.h):
lane 0 of every wave leaves the 100 MHz wall clock at its phase boundaries.
usage: YDC_LIB=yadcc_amd/libydc_probe.so python tools/phase_probe.py [cfg2|cfg3|cfg4] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402

SLOTS, CHUNKS = 12, 8192
NAMES = ["entry", "start state known", "staged + rings filled", "warm-up done", "last block done",
         "results + end state stored", "hand-off published", "predecessor arrived",
         "second replay starts", "second replay done"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    L = binding.lib()
    L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    sv, tk = synth.make_config(cfg)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    for _ in range(5):
        ctx.dispatch_device(d[0], d[1], d[2], out)
    acc = []
    for _ in range(reps):
        L.ydc_debug_phase_probe(None, 0, 1)
        ctx.dispatch_device(d[0], d[1], d[2], out)
        buf = np.zeros(CHUNKS * SLOTS, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        acc.append(buf.reshape(CHUNKS, SLOTS).astype(np.int64))
    st = ctx.stats()
    K = min(st["n_chunks"], CHUNKS)
    print("%s: %d requests, %d chunks of %d, %d classes, rounds %d; %d launches sampled; "
          "times in us (100 MHz ticks / 100)" % (cfg, st["n_tasks"], st["n_chunks"],
                                                 -(-st["n_tasks"] // max(st["n_chunks"], 1)),
                                                 st["n_classes"], st["rounds"], reps))
    T = np.stack(acc)[:, :K, :]  # reps x chunks x slots
    t0 = T[:, :, 0]
    launch0 = np.where(t0 > 0, t0, np.iinfo(np.int64).max).min(axis=1)[:, None]  # first wave in
    print("\n-- when a wave reaches a point, relative to the first wave's entry (all waves, all launches)")
    print("%-28s %8s %8s %8s %8s %8s" % ("point", "waves", "p50", "p90", "p99", "max"))
    for s in range(10):
        v = T[:, :, s]
        m = v > 0
        if not m.any():
            continue
        rel = ((v - launch0)[m]) / 100.0
        print("%-28s %8d %8.2f %8.2f %8.2f %8.2f" % (NAMES[s], m.sum() // reps, np.percentile(rel, 50),
                                                      np.percentile(rel, 90), np.percentile(rel, 99), rel.max()))
    print("\n-- duration of a wave's phases (per wave: stamp - previous stamp)")
    print("%-44s %8s %8s %8s %8s" % ("phase", "p50", "p90", "p99", "max"))
    pairs = [(0, 1, "level guesses (entry -> start state)"), (1, 2, "stage requests + fill rings"),
             (2, 3, "warm-up requests"), (3, 4, "the chunk's requests (after warm-up)"),
             (2, 4, "the chunk's requests (no warm-up: chunk 0)"), (4, 5, "store results + end state"),
             (5, 6, "publish hand-off"), (6, 7, "wait for the predecessor"), (8, 9, "second replay")]
    for a, b, name in pairs:
        va, vb = T[:, :, a], T[:, :, b]
        m = (va > 0) & (vb > 0)
        if a == 2 and b == 4:
            m &= T[:, :, 3] == 0
        if not m.any():
            continue
        dur = (vb - va)[m] / 100.0
        print("%-44s %8.2f %8.2f %8.2f %8.2f   (%d waves)" % (name, np.percentile(dur, 50), np.percentile(dur, 90),
                                                           np.percentile(dur, 99), dur.max(), m.sum() // reps))
    # Inside the block loop of the first replay (warm-up included): where the time goes.
    tu, lp = T[:, :, 10], T[:, :, 11]
    m = (T[:, :, 2] > 0) & (T[:, :, 4] > 0)
    if m.any() and (lp[m] != 0).any():
        blocks = (T[:, :, 4] - T[:, :, 2])[m] / 100.0
        t_top, n_top = (tu[m] & 0xFFFFFFFF) / 100.0, tu[m] >> 32
        t_loop, n_gen = (lp[m] & 0xFFFFFFFF) / 100.0, lp[m] >> 32
        print("\n-- inside the block loop of a chunk's first replay (per wave, p50 / p90)")
        for name, v in (("all blocks (warm-up + chunk)", blocks), ("  fast loop (asm)", t_loop),
                        ("  ring top-ups", t_top), ("  the rest (checkpoints, staging, general steps, glue)",
                                                 blocks - t_loop - t_top)):
            print("%-56s %8.2f %8.2f us" % (name, np.percentile(v, 50), np.percentile(v, 90)))
        print("%-56s %8.1f %8.1f" % ("  calls of the fast loop", np.percentile(n_top, 50), np.percentile(n_top, 90)))
        print("%-56s %8.1f %8.1f" % ("  general steps", np.percentile(n_gen, 50), np.percentile(n_gen, 90)))
    last = np.maximum.reduce([T[:, :, s] for s in range(10)]).max(axis=1)
    print("\nlast stamp of a launch - first entry: p50 %.2f us, max %.2f us" % (
        np.percentile((last - launch0[:, 0]) / 100.0, 50), ((last - launch0[:, 0]) / 100.0).max()))
    ctx.set_profiling(True)
    ctx.dispatch_device(d[0], d[1], d[2], out)
    print("kernel profile (HIP events, this build):", ctx.kernel_profile())


if __name__ == "__main__":
    main()
