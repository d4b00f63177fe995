"""Developer aid: one seeded case through libydc.so under a few switches, first mismatch against the oracle."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from oracle import oraclebind as O
    from tests import cases
    from yadcc_amd import binding, pack
    kw = eval(sys.argv[2])
    sv, tk = cases.random_case(**kw)
    want, _, _ = O.dispatch(sv, tk, "sorted")
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    got, _, _ = c.dispatch(tk)
    bad = np.nonzero(got != want)[0]
    st = c.stats()
    print("   mismatches %d first %s classes %d chunks %d rounds %d" % (
        bad.size, bad[0] if bad.size else None, st["n_classes"], st["n_chunks"], st["rounds"]))
    sys.exit(0)
base = dict(seed=91, n_tasks=40_000, n_servants=700, n_envs=1, oversubscribed=True, unknown_env_frac=0.01, self_frac=0.25)
variants = [
    ("as in the test", {"YDC_CHUNK_SIZE": "512"}, {}),
    # (a library built from an earlier commit, if there is one at build/libydc_old.so)
    ("old library", {"YDC_CHUNK_SIZE": "512", "YDC_LIB": os.path.join(ROOT, "build/libydc_old.so")}, {}),
    ("no dense", {"YDC_CHUNK_SIZE": "512", "YDC_DENSE": "0"}, {}),
    ("chunk 64", {"YDC_CHUNK_SIZE": "64"}, {}),
    ("chunk 128", {"YDC_CHUNK_SIZE": "128"}, {}),
    ("chunk 256", {"YDC_CHUNK_SIZE": "256"}, {}),
    ("default chunks", {}, {}),
    ("no fuse", {"YDC_CHUNK_SIZE": "512", "YDC_FUSE_PASSES": "0"}, {}),
    ("no binsort", {"YDC_CHUNK_SIZE": "512", "YDC_BINSORT": "0"}, {}),
    ("no level tab", {"YDC_CHUNK_SIZE": "512", "YDC_LEVEL_TAB": "0"}, {}),
    ("no self", {"YDC_CHUNK_SIZE": "512"}, {"self_frac": 0.0}),
    ("no unknown", {"YDC_CHUNK_SIZE": "512"}, {"unknown_env_frac": 0.0}),
    ("not oversubscribed", {"YDC_CHUNK_SIZE": "512"}, {"oversubscribed": False}),
    ("4 envs", {"YDC_CHUNK_SIZE": "512"}, {"n_envs": 4}),
]
for name, env, over in variants:
    if "YDC_LIB" in env and not os.path.exists(env["YDC_LIB"]):
        continue
    kw = dict(base, **over)
    print(name, env, over, flush=True)
    subprocess.run([sys.executable, __file__, "child", repr(kw)], env=dict(os.environ, **env))
