#!/usr/bin/env python3
"""When the workgroups of k_slot_gen (radix path: slot tiles and request classification in one
launch) start and finish, from the stamps of the measurement build (`make probe`).
usage: python tools/gen_probe.py [cfg3|cfg4] [reps]   (YDC_SPLIT_GEN=1: the two halves apart)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("YDC_LIB", os.path.join(ROOT, "yadcc_amd", "libydc_probe.so"))
from yadcc_amd import binding, pack, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
    rows = []
    for _ in range(reps):
        L.ydc_debug_phase_probe(None, 0, 1)
        ctx.dispatch_device(d[0], d[1], d[2], out)
        buf = np.zeros(8192 * 12, np.uint64)
        L.ydc_debug_phase_probe(buf.ctypes.data, buf.size, 0)
        b = buf.astype(np.int64)
        grid, gen_blocks, stride = int(b[39990]), int(b[39991]), max(1, int(b[39992]))
        st = b[40000:40000 + 4800].reshape(2400, 2)
        n = -(-grid // stride)
        blk = np.arange(n) * stride
        rows.append((st[:n], blk < gen_blocks))
    # (with YDC_SPLIT_GEN=1 the last launch stamped is the classification: gen_blocks == 0)
    print("%s: k_slot_gen grid %d = %d slot tiles + %d request blocks, every %d-th workgroup stamped; "
          "us after the first stamped start" % (cfg, grid, gen_blocks, grid - gen_blocks, stride))
    for name, want in (("slot tiles", True), ("request blocks", False)):
        S, E = [], []
        for st, is_gen in rows:
            m = (is_gen == want) & (st[:, 0] > 0) & (st[:, 1] > 0)
            if not m.any():
                continue
            t0 = st[st[:, 0] > 0, 0].min()
            S.append((st[m, 0] - t0) / 100.0)
            E.append((st[m, 1] - t0) / 100.0)
        if not S:
            continue
        S, E = np.concatenate(S), np.concatenate(E)
        D = E - S
        print("  %-15s start p50 %7.2f p99 %7.2f | end p50 %7.2f p99 %7.2f max %7.2f | duration p50 %6.2f p90 %6.2f max %6.2f" % (
            name, np.percentile(S, 50), np.percentile(S, 99), np.percentile(E, 50), np.percentile(E, 99),
            E.max(), np.percentile(D, 50), np.percentile(D, 90), D.max()))


if __name__ == "__main__":
    main()
