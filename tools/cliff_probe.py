#!/usr/bin/env python3
"""The slow corner of DESIGN.md 9.7 (tests/tools/fuzz_parity.py seed 72157: three servants offering
140k slots, one class, a tenth of 120k requests from the servants' own hosts): time per batch and
per kernel for several `walk_after`, and — with `probe` (the measurement build, `make probe`) — where
the walking wave spends its time.   usage: python tools/cliff_probe.py [probe]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = "probe" in sys.argv[1:]
if PROBE:
    os.environ["YDC_LIB"] = os.path.join(ROOT, "yadcc_amd", "libydc_probe.so")
from tests import cases  # noqa: E402
from yadcc_amd import binding, pack  # noqa: E402


def main():
    sv, tk = cases.random_case(seed=72157, n_tasks=120000, n_servants=3, n_envs=1, self_frac=0.1,
                               unknown_env_frac=0.0, min_version_20_frac=0.0, initial_running=True)
    for wa in ((3, 6, 12) if not PROBE else (12,)):
        os.environ["YDC_TUNE"] = "walk_after=%d" % wa
        c = binding.Context(device=0)
        c.upload_servants(pack.to_abi_columns(sv))
        c.dispatch(tk)
        t0 = time.perf_counter()
        c.dispatch(tk)
        dt = time.perf_counter() - t0
        st = c.stats()
        c.set_profiling(True)
        if PROBE:
            L = binding.lib()
            L.ydc_debug_phase_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
            L.ydc_debug_phase_probe(None, 0, 1)
        c.dispatch(tk)
        kp = c.kernel_profile()
        print("walk_after %2d: %d passes, %.1f ms per batch; k_match_pass %d launches, %.1f ms"
              % (wa, st["rounds"], dt * 1e3, kp["k_match_pass"][0], kp["k_match_pass"][1]))
        if PROBE:
            buf = np.zeros(12, np.uint64)
            L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
            b = [int(x) for x in buf]
            lo = lambda v: (v & 0xFFFFFFFF) / 100.0  # noqa: E731
            print("   the walking wave: %.0f us from its start to its last chunk's end; %d calls of the fast loop "
                  "(%.0f us in it, %.0f us topping rings up before it), %d general steps (%.0f us)"
                  % ((b[9] - b[5]) / 100.0, b[6] >> 32, lo(b[7]), lo(b[6]), b[7] >> 32, lo(b[8])))
        c.close()


if __name__ == "__main__":
    main()
