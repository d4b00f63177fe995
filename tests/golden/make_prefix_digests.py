"""Generates tests/golden/ref_cfg{3,4}_prefix_digests.npz: the VERBATIM reference's placement of a
long prefix of the full-size batches of BASELINE.json configs[2] / configs[3] — cfg3's first
400k requests (which include the dedicated-tier boundary around request 337k, where the matching
passes have to follow a chain of ~2000 requests), cfg4's first 200k — stored as one
order-sensitive digest per block of 1000 requests (yadcc_amd.synth.placement_hash), the grant
count per block, and a checksum of the inputs. A sequential batch's prefix is the prefix batch,
so the first placements of any full-size run must reproduce them. Run in the build container
(needs /root/reference): python tests/golden/make_prefix_digests.py   (~2 min)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refbind as R  # noqa: E402
from yadcc_amd import synth  # noqa: E402

BLOCK = 1000
SPECS = {"cfg3": 400_000, "cfg4": 200_000}


def input_checksum(sv, tk, n):
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    return h.hexdigest()


def block_digests(idx):
    n = len(idx) // BLOCK
    return (np.array([synth.placement_hash(idx[b * BLOCK:(b + 1) * BLOCK]) for b in range(n)], np.uint64),
            np.array([int((idx[b * BLOCK:(b + 1) * BLOCK] < R.IDX_ENV_NOT_FOUND).sum()) for b in range(n)],
                     np.uint32))


if __name__ == "__main__":
    assert R.available(), "oracle/_ref is not built (needs /root/reference)"
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for cfg, n in SPECS.items():
        sv, tk = synth.make_config(cfg)
        head = {k: v[:n] for k, v in tk.items()}
        d = R.RefDispatcher()
        d.load_servants(sv)
        idx, _, secs, _ = d.dispatch_batch(head)
        d.close()
        dig, granted = block_digests(idx)
        np.savez_compressed(os.path.join(out_dir, "ref_%s_prefix_digests.npz" % cfg), config=cfg, prefix=n,
                            block=BLOCK, input_sha256=input_checksum(sv, tk, n), digest=dig, granted=granted)
        print(cfg, "first", n, "requests, ref %.1f s," % secs, int(granted.sum()), "granted")
