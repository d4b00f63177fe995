#!/usr/bin/env python3
"""Per-kernel times (HIP events, one profiled batch after three warm ones) of a configuration's
batch, optionally with fewer requests on the same registry.
usage: python tools/kernel_times.py cfg4 [n_tasks]      (environment: YDC_TUNE / YDC_<KEY>)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yadcc_amd import binding, pack, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    sv, tk = synth.make_config(cfg)
    if len(sys.argv) > 2:
        n = int(sys.argv[2])
        tk = {k: v[:n] for k, v in tk.items()}
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(np.ascontiguousarray(tk[k])) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    for _ in range(3):
        ctx.dispatch_device(d[0], d[1], d[2], out)
    ctx.set_profiling(True)
    acc = {}
    for _ in range(5):
        ctx.dispatch_device(d[0], d[1], d[2], out)
        for k, v in ctx.kernel_profile().items():
            acc.setdefault(k, []).append(v[1] * 1e3)
    print("%s, %d requests x %d servants:" % (cfg, len(tk["env_id"]), len(sv["version"])),
          "  ".join("%s %.1f" % (k[2:], float(np.median(v))) for k, v in acc.items()))


if __name__ == "__main__":
    main()
