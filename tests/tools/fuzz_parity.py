#!/usr/bin/env python3
"""Developer tool: randomised differential run of the HIP path against the oracle.

Draws registries and batches from the seeded families of tests/cases.py with random shapes
(1..8 digests — one or two half of the time: few classes, long class lists —, shared hosts,
oversubscription, initial running_tasks, unknown digests, self requests, forced chunk and ring
sizes, batches committed in two halves) for a given number of seconds and reports every mismatch.
    python tests/tools/fuzz_parity.py [seconds=60] [first_seed=1000] [max_cases=0]
max_cases > 0 ends the run after that many cases (tests/test_fuzz_gpu.py: a bounded run inside
`pytest -m gpu`, same seeds every time). Needs the GPU; the oracle is the checker (test
infrastructure)."""
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
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    t_end = time.time() + budget
    n_cases = n_bad = 0
    ctxs = {}
    shapes = {}
    while time.time() < t_end and not (max_cases and n_cases >= max_cases):
        rng = np.random.default_rng(seed)
        kw = dict(seed=seed,
                  n_tasks=int(rng.choice([1, 63, 64, 65, 700, 5000, 30000, 120000])),
                  n_servants=int(rng.choice([1, 3, 40, 300, 1500, 5000])),
                  n_envs=int(rng.integers(1, 3)) if rng.random() < 0.5 else int(rng.integers(1, 9)),
                  self_frac=float(rng.choice([0.0, 0.1, 0.5])),
                  unknown_env_frac=float(rng.choice([0.0, 0.01])),
                  min_version_20_frac=float(rng.choice([0.0, 0.5, 1.0])))
        if rng.random() < 0.25:
            kw["oversubscribed"] = True
        if rng.random() < 0.2:
            kw["shared_ip_frac"] = 0.2
        if rng.random() < 0.3:
            kw["initial_running"] = True
        if rng.random() < 0.2:
            kw["disjoint_envs"] = True
        chunk = int(rng.choice([0, 0, 64, 128, 512, 1024, 4096]))
        rings = int(rng.choice([0, 0, 256, 1024]))
        halves = rng.random() < 0.3
        sv, tk = cases.random_case(**kw)
        if rng.random() < 0.3:  # a few versions more: more classes
            sv["version"] = (19 + rng.integers(0, 4, len(sv["version"]))).astype(np.uint32)
        if (chunk, rings) not in ctxs:
            for name, v in (("YDC_CHUNK_SIZE", chunk), ("YDC_RING_TOTAL", rings)):
                if v:
                    os.environ[name] = str(v)
                else:
                    os.environ.pop(name, None)
            ctxs[(chunk, rings)] = binding.Context(device=0)
        ctx = ctxs[(chunk, rings)]
        t_case = time.time()
        if os.environ.get("YDC_FUZZ_VERBOSE"):
            print("case seed %d chunk %d rings %d halves %s kw %s" % (seed, chunk, rings, halves, kw), flush=True)
        want, wutil, wrun = O.dispatch(sv, tk, "sorted")
        t_oracle = time.time() - t_case
        ctx.upload_servants(pack.to_abi_columns(sv))
        n = len(tk["env_id"])
        if halves and n > 1:
            cut = int(rng.integers(1, n))
            a, ua, _ = ctx.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
            b, ub, grun = ctx.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
            got, gutil = np.concatenate([a, b]), np.concatenate([ua, ub])
        else:
            got, gutil, grun = ctx.dispatch(tk)
        st = ctx.stats()
        key = (st["n_classes"] > 64, st["n_classes"] > 4, st["key_bits"])
        shapes[key] = shapes.get(key, 0) + 1
        ok = np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
        n_cases += 1
        if time.time() - t_case > 2.0:
            print("SLOW seed %d: %.1f s (oracle %.1f s) chunk %d rings %d kw %s stats %s" % (
                seed, time.time() - t_case, t_oracle, chunk, rings, kw, st), flush=True)
        if not ok:
            n_bad += 1
            bad = np.nonzero(got != want)[0]
            print("MISMATCH seed %d chunk %d rings %d halves %s kw %s: %d requests differ (first %s) stats %s" % (
                seed, chunk, rings, halves, kw, bad.size, bad[:3], st), flush=True)
        seed += 1
    print("fuzz: %d cases, %d mismatches; shapes (>64 classes, >4 classes, key bits): %s" % (
        n_cases, n_bad, shapes))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
