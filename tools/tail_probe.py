#!/usr/bin/env python3
"""Which waves of k_match_pass are the slow ones? Per-wave duration of the chunk's requests
(stamps of the measurement build, like tools/phase_probe.py) against the chunk index: by
chunk mod 8 (the XCD a workgroup lands on), in runs of consecutive chunks, against the counts
of general steps / fast-loop calls, and repeatability across launches.
usage: YDC_LIB=yadcc_amd/libydc_probe.so python tools/tail_probe.py [cfg4] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402

SLOTS, CHUNKS = 12, 8192


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = binding.lib()
    L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    sv, tk = synth.make_config(cfg)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    for _ in range(3):
        ctx.dispatch_device(d[0], d[1], d[2], out)
    acc = []
    for _ in range(reps):
        L.ydc_debug_phase_probe(None, 0, 1)
        ctx.dispatch_device(d[0], d[1], d[2], out)
        buf = np.zeros(CHUNKS * SLOTS, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        acc.append(buf.reshape(CHUNKS, SLOTS).astype(np.int64))
    K = min(ctx.stats()["n_chunks"], CHUNKS)
    T = np.stack(acc)[:, :K, :]
    dur = (T[:, :, 4] - T[:, :, 2]) / 100.0  # staged -> last block done, us
    entry = (T[:, :, 0] - T[:, :, 0].min(axis=1, keepdims=True)) / 100.0
    loop = (T[:, :, 11] & 0xFFFFFFFF) / 100.0
    gen = T[:, :, 11] >> 32
    calls = T[:, :, 10] >> 32
    print("%s: %d chunks, %d launches; block-loop duration p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (
        cfg, K, reps, *np.percentile(dur, [10, 50, 90]), dur.max()))
    print("\nby chunk mod 8 (XCD of the workgroup): p50 of duration")
    print("  ", " ".join("%7.1f" % np.median(dur[:, x::8]) for x in range(8)))
    print("by chunk mod 16:", " ".join("%6.1f" % np.median(dur[:, x::16]) for x in range(16)))
    print("\nrepeatability: correlation of a chunk's duration between launches: %.3f" % np.corrcoef(dur[0], dur[-1])[0, 1])
    m = np.median(dur, axis=0)  # per chunk over launches
    print("per-chunk median over launches: p10 %.1f p50 %.1f p90 %.1f" % tuple(np.percentile(m, [10, 50, 90])))
    print("correlation of the per-chunk median with: general steps %.3f, fast-loop calls %.3f, entry time %.3f" % (
        np.corrcoef(m, np.median(gen, axis=0))[0, 1], np.corrcoef(m, np.median(calls, axis=0))[0, 1],
        np.corrcoef(m, np.median(entry, axis=0))[0, 1]))
    slow = m > np.percentile(m, 85)
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], slow.astype(int), [0]])))) [::2]
    print("slow chunks (top 15 %%): %d, in %d runs of consecutive chunks, longest %d, first slow chunk %d, last %d" % (
        slow.sum(), len(runs), runs.max() if len(runs) else 0, np.flatnonzero(slow)[0], np.flatnonzero(slow)[-1]))
    q = K // 16
    print("median duration by sixteenth of the batch:", " ".join("%6.1f" % np.median(m[i * q:(i + 1) * q]) for i in range(16)))
    print("median general steps by sixteenth:        ", " ".join("%6.0f" % np.median(np.median(gen, axis=0)[i * q:(i + 1) * q]) for i in range(16)))
    print("median fast-loop time by sixteenth:       ", " ".join("%6.1f" % np.median(np.median(loop, axis=0)[i * q:(i + 1) * q]) for i in range(16)))


if __name__ == "__main__":
    main()
