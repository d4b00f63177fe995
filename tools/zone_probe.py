#!/usr/bin/env python3
"""The walk of the dedicated tier's end (zone_guess.h) on variants of cfg3 — other seeds of the
registry, disjoint environment partitions, shorter batches: rounds, served chunks and time per
batch with the walk (zone_guess=2), without (0) and decided by the library (1, the default), placement
against the oracle.
usage: python tools/zone_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oraclebind as O  # noqa: E402
from yadcc_amd import binding, pack, synth  # noqa: E402


def run(sv, tk, tune):
    os.environ["YDC_TUNE"] = tune
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    out = DA(len(tk["env_id"]), np.uint32)
    for _ in range(6):
        c.dispatch_device(d[0], d[1], d[2], out)
    t0 = time.perf_counter()
    for _ in range(20):
        c.dispatch_device(d[0], d[1], d[2], out)
    dt = (time.perf_counter() - t0) / 20
    st = c.stats()
    got = out.numpy()
    c.close()
    return got, st, dt


def main():
    variants = [("cfg3", {}), ("cfg3 seed 2", {"seed": 2}), ("cfg3 seed 3", {"seed": 3}), ("cfg3 seed 4", {"seed": 4}),
                ("cfg3 disjoint digests", {"disjoint_envs": True}),
                ("cfg3, 5 % of the servants on shared hosts", {"shared_ip_frac": 0.05}),
                ("cfg3 registry, 600k requests", {"n_tasks": 600_000}), ("cfg3 with 6 digests", {"n_envs": 6})]
    for name, kw in variants:
        sv, tk = synth.make_config("cfg3", **kw)
        want = O.dispatch(sv, tk, "sorted")[0]
        a, sa, ta = run(sv, tk, "zone_guess=2")
        b, sb, tb = run(sv, tk, "zone_guess=0")
        d, sd, td = run(sv, tk, "zone_guess=1")
        print("%-44s %2d classes, %4d chunks | walk: %d chunks served, %d rounds, %.3f ms | without: %d rounds, %.3f ms | "
              "by itself: %s, %.3f ms | %s"
              % (name, sa["n_classes"], sa["n_chunks"], sa["zone_rows"], sa["rounds"], ta * 1e3, sb["rounds"], tb * 1e3,
                 "walk" if sd["zone_rows"] else "no walk", td * 1e3,
                 "same as the oracle" if all(np.array_equal(x, want) for x in (a, b, d)) else "MISMATCH"))


if __name__ == "__main__":
    main()
