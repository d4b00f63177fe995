import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oraclebind as O
from tests import cases
from tests.test_sharded_gpu import make_group, sharded_run
os.environ["YDC_SHARD_MARGIN"] = sys.argv[1] if len(sys.argv) > 1 else "2000"
os.environ["YDC_DEBUG_SIM"] = "1"
kw = {'seed': 2009, 'n_tasks': 150000, 'n_servants': 5, 'n_envs': 3, 'self_frac': 0.1, 'unknown_env_frac': 0.01, 'initial_running': True}
sv, tk = cases.random_case(**kw)
print("nproc", sv["num_processors"], "max_tasks", sv["max_tasks"], "running", sv["running_tasks"], "env", sv["env_mask"], flush=True)
from yadcc_amd import binding, pack
os.environ.pop("YDC_DEBUG_SIM", None)
ctx = binding.Context(device=0)
ctx.upload_servants(pack.to_abi_columns(sv))
got, _, _ = ctx.dispatch(tk)
want = O.dispatch(sv, tk, "sorted")[0]
print("single GPU: ok", np.array_equal(got, want), ctx.stats())
