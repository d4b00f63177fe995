#!/usr/bin/env python3
"""Where the walk of the wide kernel (k_sim_wide, walk != 0) spends its time: accumulated
100 MHz wall-clock ticks per part, from the measurement build (`make probe`).
usage: YDC_LIB=yadcc_amd/libydc_probe.so python tools/walk_probe.py [digests] [requests]
       ... tools/walk_probe.py groups [digests] [requests]: the walk 64 requests at a time
       (k_walk_groups) — ticks per part of an iteration, iterations, losers, marking rounds"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402


def groups():
    digests = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    L = binding.lib()
    L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    sv, tk = synth.make_config("cfg2", n_envs=digests, n_tasks=n)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    ctx.dispatch_device(d[0], d[1], d[2], out)
    L.ydc_debug_phase_probe(None, 0, 1)
    ctx.dispatch_device(d[0], d[1], d[2], out)
    buf = np.zeros(32, np.uint64)
    L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
    b = [float(x) for x in buf[16:32]]
    st = ctx.stats()
    tot = b[6] / 100.0
    blocks = (n + 63) // 64
    print("k_walk_groups: %d requests, %d classes; %.0f us in all (with the stamps); %d blocks, %d iterations "
          "(%.2f per block), %d of them general steps" % (st["n_tasks"], st["n_classes"], tot, blocks, b[7], b[7] / blocks, b[8]))
    print("   per iteration: %.1f unresolved lanes at its start, %.1f commits, %.1f losers of a claim, %.1f blocked by a mark, "
          "%.2f marking rounds" % (b[13] / max(b[7], 1), b[10] / max(b[7] - b[8], 1), b[11] / max(b[7] - b[8], 1),
                                   b[12] / max(b[7] - b[8], 1), b[9] / max(b[7] - b[8], 1)))
    fast = max(b[7] - b[8], 1)
    for nm, v, per in (("block start (rows into registers)", b[0], blocks), ("scan of the rows", b[1], b[7]),
                       ("general steps", b[2], max(b[8], 1)), ("claim", b[3], fast), ("marking rounds", b[4], fast),
                       ("flush + commit", b[5], fast)):
        print("   %-36s %9.0f us (%4.1f %%)  %7.3f us each" % (nm, v / 100.0, 100.0 * v / max(b[6], 1), v / 100.0 / per))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "groups":
        return groups()
    digests = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    L = binding.lib()
    L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    sv, tk = synth.make_config("cfg2", n_envs=digests, n_tasks=n)
    os.environ["YDC_WALK_PREFETCH"] = "1"
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    for mode in ("prefetch waves", "lone wave"):
        if mode == "lone wave":
            ctx.close()
            os.environ["YDC_WALK_PREFETCH"] = "0"
            ctx = binding.Context(device=0)
            ctx.upload_servants(pack.to_abi_columns(sv))
        ctx.dispatch_device(d[0], d[1], d[2], out)
        L.ydc_debug_phase_probe(None, 0, 1)
        ctx.set_profiling(True)
        ctx.dispatch_device(d[0], d[1], d[2], out)
        buf = np.zeros(16, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        st = ctx.stats()
        names = ["stage a block (self columns, masks)", "scan + minimum", "consume (ring / fetch)", "general steps",
                 "chunk end (end state, guess)"]
        tot = float(buf[7]) / 100.0
        print("%s: %d requests, %d classes, rounds %d; walk %.0f us in all, %d fast / %d general requests"
              % (mode, st["n_tasks"], st["n_classes"], st["rounds"], tot, int(buf[5]), int(buf[6])))
        for i, nm in enumerate(names):
            print("   %-40s %10.0f us  (%4.1f %%)" % (nm, float(buf[i]) / 100.0, 100.0 * float(buf[i]) / max(float(buf[7]), 1)))
        print("   kernel profile:", {k: round(v[1] * 1e3) for k, v in ctx.kernel_profile().items() if "wide" in k})


if __name__ == "__main__":
    main()
