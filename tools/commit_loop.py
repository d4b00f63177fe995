#!/usr/bin/env python3
"""The dependent loop of a scheduler on cfg2: every batch COMMITs, its grants are released again
before the next one (ydc_release_slots_device: indexes already in HBM). ms per step for the
release kernel in both forms (YDC_TUNE=release_counted=0 / 1).   usage: python tools/commit_loop.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yadcc_amd import binding, pack, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sv, tk = synth.make_config("cfg2")
    DA = binding.DeviceArray
    for tune in ("release_counted=1", "release_counted=0"):
        os.environ["YDC_TUNE"] = tune
        c = binding.Context(device=0)
        c.upload_servants(pack.to_abi_columns(sv))
        d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
        out = DA(len(tk["env_id"]), np.uint32)
        before = c.get_running().copy()
        for i in range(steps + 20):
            if i == 20:
                c.synchronize()
                t0 = time.perf_counter()
            c.dispatch_device(d[0], d[1], d[2], out, commit=True)
            c.release_slots_device(out)
        c.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ok = np.array_equal(c.get_running(), before)
        c.close()
        print("%-20s %.4f ms per step (COMMIT + release of the batch's 100000 grants from HBM), registry restored: %s"
              % (tune, dt * 1e3, ok))


if __name__ == "__main__":
    main()
