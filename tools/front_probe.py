#!/usr/bin/env python3
"""When the workgroups of k_front_bins (by role) and the phases of k_bin_sort finish, from the
stamps of the measurement build (`make probe`). usage: python tools/front_probe.py [cfg2] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402


def pct(v, q):
    return float(np.percentile(v, q)) if len(v) else float("nan")


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
    n_req = -(-len(tk["env_id"]) // 256)
    front, sort = [], []
    for _ in range(reps):
        L.ydc_debug_phase_probe(None, 0, 1)
        ctx.dispatch_device(d[0], d[1], d[2], out)
        buf = np.zeros(8192 * 12, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        b = buf.astype(np.int64)
        front.append(b[40000:40000 + 4800].reshape(2400, 2))
        sort.append(b[46000:46000 + 12600].reshape(2100, 6))
    F = np.stack(front)
    used = F[0, :, 0] > 0
    n_wg = int(used.sum())
    # (slot tiles first, then the bin boundaries — a power of two, more of them than tiles —, then requests)
    n_bins = 1 << int(np.floor(np.log2(n_wg - n_req - 1)))
    n_tiles = n_wg - n_bins - n_req
    print("%s: k_front_bins %d workgroups = %d bin boundaries + %d slot tiles + %d request blocks; us after the first workgroup's start"
          % (cfg, n_wg, n_bins, n_tiles, n_req))
    t0 = np.where(F[:, :, 0] > 0, F[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]
    for name, lo, hi in (("slot tiles", 0, n_tiles), ("bin boundaries", n_tiles, n_bins + n_tiles),
                         ("request blocks", n_bins + n_tiles, n_wg)):
        st = (F[:, lo:hi, 0] - t0) / 100.0
        en = (F[:, lo:hi, 1] - t0) / 100.0
        print("  %-16s start p50 %6.2f max %6.2f | end p50 %6.2f p99 %6.2f max %6.2f | duration p50 %6.2f max %6.2f" % (
            name, pct(st, 50), st.max(), pct(en, 50), pct(en, 99), en.max(), pct(en - st, 50), (en - st).max()))
    S = np.stack(sort)
    used = S[0, :, 0] > 0
    nb = int(used.sum())
    s0 = np.where(S[:, :, 0] > 0, S[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None]
    print("k_bin_sort: %d workgroups; us after the first workgroup's start" % nb)
    names = ["start", "run table read + tile starts", "records staged in LDS", "counting passes done", "places written"]
    for i, nm in enumerate(names):
        v = S[:, :nb, i]
        m = v > 0
        rel = ((v - s0)[m]) / 100.0
        print("  %-32s p50 %6.2f p99 %6.2f max %6.2f  (%d workgroups)" % (nm, pct(rel, 50), pct(rel, 99), rel.max(), m.sum() // reps))
    ctx.set_profiling(True)
    ctx.dispatch_device(d[0], d[1], d[2], out)
    print("kernel profile (events, this build):", ctx.kernel_profile())


if __name__ == "__main__":
    main()
