#!/usr/bin/env python3
"""Where a small-batch launch (k_tick, yadcc_amd/csrc/tick_kernel.h) spends its time: phase stamps
of the measurement build (`make probe` -> yadcc_amd/libydc_probe.so), thread 0's 100 MHz wall clock.
usage: python tools/tick_probe.py [servants] [requests] [releases] [reps] [distinct]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402

NAMES = {0: "entry", 1: "argument head in SGPRs", 2: "columns loaded, keys computed", 3: "staged arguments visible",
         4: "deltas applied", 5: "first signature: masks, eligibility", 6: "running_tasks written back",
         7: "system fence + barrier"}


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    n_rel = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    L = binding.lib()
    L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    sv = synth.make_servants(S, n_tasks_hint=100 * S, n_envs=4, seed=3)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    tk = synth.make_tasks(n, sv, n_envs=4, seed=5, self_frac=0.0)
    distinct = len(sys.argv) > 5 and sys.argv[5] == "distinct"  # (default: one RPC, one signature)
    if not distinct:
        for k in tk:
            tk[k][:] = tk[k][0]
    acc, wall = [], []
    held = []
    for r in range(reps + 20):
        rel = [held.pop() for _ in range(min(n_rel, len(held)))]
        L.ydc_debug_phase_probe(None, 0, 1)
        t0 = time.perf_counter()
        got, _ = ctx.dispatch_tick(tk, release_idx=rel)
        t1 = time.perf_counter()
        held.extend(int(s) for s in got if s < 0xFFFFFFFE)
        while len(held) > 4 * max(n, n_rel):
            ctx.dispatch_tick({k: v[:0] for k, v in tk.items()}, release_idx=[held.pop() for _ in range(n)])
        buf = np.zeros(32, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        if r >= 20:
            acc.append(buf.astype(np.int64))
            wall.append((t1 - t0) * 1e6)
    T = np.stack(acc)
    assert ctx.stats()["small_batch"] == 1
    print("k_tick on %d servants, %d requests of %s, %d released grants per call; %d calls; "
          "microseconds after the kernel's entry (p50)" % (S, n, "their own signatures" if distinct else "one signature", n_rel, reps))
    rel = (T - T[:, :1]) / 100.0
    order = [0, 1, 2, 3, 4] + ([5] + list(range(8, 8 + min(n, 16))) if not T[:, 27].any() else []) + [6, 7]
    prev = 0.0
    for s in order:
        if not T[:, s].all():  # (a stamp this variant of the kernel does not set)
            continue
        v = np.median(rel[:, s])
        print("  %-40s %8.2f  (+%.2f)" % (NAMES.get(s, "pick %d done" % (s - 8)), v, v - prev))
        prev = v
    if n > 2 and not T[:, 27].any():
        for a, b, what in ((9, 24, "pick 2: candidates rescanned (the thread whose servant won pick 1)"),
                           (24, 25, "pick 2: workgroup reduction"), (25, 26, "pick 2: own-host flag (+ second reduction)"),
                           (26, 10, "pick 2: winner's state, key, results")):
            print("  %-70s %6.2f" % (what, np.median((T[:, b] - T[:, a]) / 100.0)))
    if T[:, 27].any():
        print("  merge: %d round(s), %d placed in the first; first round: lists built + visible at %.2f, merged at %.2f, "
              "applied at %.2f us" % (np.median(T[:, 30]), np.median(T[:, 31]), np.median(rel[:, 27]),
                                      np.median(rel[:, 28]), np.median(rel[:, 29])))
    print("host: python call p50 %.1f us (ctypes + numpy marshalling included)" % np.median(wall))


if __name__ == "__main__":
    main()
