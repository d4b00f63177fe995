import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oraclebind as O
from tests import cases
from tests.test_sharded_gpu import make_group, sharded_run
sv, tk = cases.random_case(seed=65, n_tasks=9000, n_servants=3000, n_envs=10)
sv["version"] = (20 + np.arange(3000) % 3).astype(np.uint32)
n = len(tk["env_id"])
want, wutil, wrun = O.dispatch(sv, tk, "sorted")
for it in range(8):
    ctxs = make_group(3, sv)
    res = sharded_run(ctxs, sv, tk, [0, n // 5, n // 5, n], commit=True)
    got = np.concatenate([r[0] for r in res])
    bad = np.nonzero(got != want)[0]
    print("iter", it, "mismatches", bad.size, "first", bad[:5], got[bad[:5]], want[bad[:5]], "rounds", [r[3]["rounds"] for r in res], "run ok", [bool(np.array_equal(r[2], wrun)) for r in res])
    [c.close() for c in ctxs]
