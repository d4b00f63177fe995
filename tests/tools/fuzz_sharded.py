#!/usr/bin/env python3
"""Developer tool: randomised differential run of the SHARDED path (several ranks as contexts
on one GPU, tests/test_sharded_gpu.py's transport) against the oracle: random registries,
rank counts, uneven cuts with empty slices, oversubscription, shared hosts, disjoint
partitions — with the sharded sort (key windows) and its fallback both in play.
    python tests/tools/fuzz_sharded.py [seconds=60] [first_seed=2000]
Needs the GPU; the oracle is the checker (test infrastructure)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oraclebind as O  # noqa: E402
from tests import cases  # noqa: E402
from tests.test_sharded_gpu import make_group, sharded_run  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    t_end = time.time() + budget
    n_cases = n_bad = windowed = misses = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        kw = dict(seed=seed,
                  n_tasks=int(rng.choice([300, 5000, 40000, 150000])),
                  n_servants=int(rng.choice([60, 400, 2500])),
                  n_envs=int(rng.integers(1, 7)),
                  self_frac=float(rng.choice([0.0, 0.1, 0.4])),
                  unknown_env_frac=float(rng.choice([0.0, 0.01])))
        if rng.random() < 0.3:
            kw["oversubscribed"] = True
        if rng.random() < 0.15:
            kw["shared_ip_frac"] = 0.2
        if rng.random() < 0.3:
            kw["initial_running"] = True
        if rng.random() < 0.15:
            kw["disjoint_envs"] = True
        # (Pools of a handful of servants with tens of thousands of slots each and heavy traffic
        # from their own hosts converge one chunk per pass — every start state has holes no guess
        # predicts; exact, but minutes over this test transport. Not a realistic shape.)
        kw["n_tasks"] = min(kw["n_tasks"], 300 * kw["n_servants"])
        margin = rng.choice(["", "", "0", "64", "2000"])
        if margin:
            os.environ["YDC_SHARD_MARGIN"] = margin
        else:
            os.environ.pop("YDC_SHARD_MARGIN", None)
        sv, tk = cases.random_case(**kw)
        n = len(tk["env_id"])
        G = int(rng.integers(2, 7))
        cuts = [0] + sorted(rng.integers(0, n + 1, G - 1).tolist()) + [n]
        if rng.random() < 0.3:
            cuts[1] = 0  # an empty first slice
        print("case seed %d G %d cuts %s margin %r kw %s" % (seed, G, cuts, margin, kw), flush=True)
        ctxs = make_group(G, sv)
        res = sharded_run(ctxs, sv, tk, cuts, commit=bool(rng.random() < 0.5))
        want, wutil, wrun = O.dispatch(sv, tk, "sorted")
        got = np.concatenate([r[0] for r in res])
        ok = np.array_equal(got, want) and all(np.array_equal(r[2], wrun) for r in res) and \
            np.array_equal(np.concatenate([r[1] for r in res]), wutil)
        st = res[0][3]
        windowed += st["shard_sort_batches"]
        misses += st["shard_sort_misses"]
        n_cases += 1
        if not ok:
            n_bad += 1
            bad = np.nonzero(got != want)[0]
            print("MISMATCH seed %d G %d cuts %s margin %r kw %s: %d requests differ (first %s)" % (
                seed, G, cuts, margin, kw, bad.size, bad[:3]), flush=True)
        [c.close() for c in ctxs]
        seed += 1
    print("fuzz_sharded: %d cases, %d mismatches; %d ran with a sharded sort, %d of them fell back" % (
        n_cases, n_bad, windowed, misses))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
