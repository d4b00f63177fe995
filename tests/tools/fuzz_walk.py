#!/usr/bin/env python3
"""Developer tool: randomised differential run of the sparse-eligibility walk (k_walk_groups:
40 .. 250 digests, about one servant class per servant) against the oracle — the shapes of
tests/test_gpu_parity.py::test_group_walk_random_sparse_pools over many more seeds, both head
layouts.   python tests/tools/fuzz_walk.py [seconds=60] [first_seed=5000]
Needs the GPU; the oracle is the checker (test infrastructure)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oraclebind as O  # noqa: E402
from tests import cases  # noqa: E402
from yadcc_amd import binding, pack  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t_end = time.time() + budget
    n_cases = n_bad = n_walked = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        kw = dict(seed=seed, n_tasks=int(rng.choice([1, 63, 64, 65, 129, 1000, 4097, 20000, 50000])),
                  n_servants=int(rng.choice([280, 300, 700, 1500, 2500])),
                  n_envs=int(rng.choice([40, 90, 150, 200, 250])),
                  self_frac=float(rng.choice([0.0, 0.1, 0.6])), unknown_env_frac=float(rng.choice([0.0, 0.01])),
                  min_version_20_frac=float(rng.choice([0.0, 0.5, 1.0])))
        if rng.random() < 0.3:
            kw["oversubscribed"] = True
        if rng.random() < 0.3:
            kw["shared_ip_frac"] = 0.25
        if rng.random() < 0.3:
            kw["initial_running"] = True
        packed = rng.random() < 0.7
        os.environ["YDC_WALK_PACKED"] = "1" if packed else "0"
        sv, tk = cases.random_case(**kw)
        if rng.random() < 0.3:  # a few versions more: more classes
            sv["version"] = (19 + rng.integers(0, 4, len(sv["version"]))).astype(np.uint32)
        n = len(tk["env_id"])
        want, wutil, wrun = O.dispatch(sv, tk, "scan" if n <= 5000 else "sorted")
        c = binding.Context(device=0)
        try:
            c.upload_servants(pack.to_abi_columns(sv))
            if rng.random() < 0.3 and n > 1:
                cut = int(rng.integers(1, n))
                a, ua, _ = c.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
                b, ub, grun = c.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
                got, gutil = np.concatenate([a, b]), np.concatenate([ua, ub])
            else:
                got, gutil, grun = c.dispatch(tk)
            st = c.stats()
        finally:
            c.close()
        n_cases += 1
        n_walked += st.get("n_classes", 0) > 256
        ok = np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
        if not ok:
            n_bad += 1
            bad = np.nonzero(got != want)[0]
            print("MISMATCH seed %d packed %d %s first bad %s classes %s" % (seed, packed, kw, bad[:5], st.get("n_classes")))
        seed += 1
    print("%d cases (%d with more than 256 classes), %d mismatches, seeds up to %d" % (n_cases, n_walked, n_bad, seed - 1))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
