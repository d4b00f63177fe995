"""Generates tests/golden/*.npz from the VERBATIM reference (oracle/_ref, i.e.
/root/reference/yadcc/scheduler/task_dispatcher.cc compiled unmodified). Run in the
build container (needs /root/reference); the .npz files are committed so the GPU box,
which has no /root/reference, can check against them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refbind as R  # noqa: E402
from tests import cases  # noqa: E402

SPECS = {
    "ref_single_env_1k_x_64": dict(seed=101, n_tasks=1000, n_servants=64, self_frac=0.2),
    "ref_four_envs_8k_x_300": dict(seed=102, n_tasks=8000, n_servants=300, n_envs=4,
                                   unknown_env_frac=0.002, self_frac=0.25),
    "ref_shared_hosts_3k_x_120": dict(seed=103, n_tasks=3000, n_servants=120, n_envs=3,
                                      shared_ip_frac=0.25, self_frac=0.4),
    "ref_oversubscribed_12k_x_150": dict(seed=104, n_tasks=12000, n_servants=150, n_envs=2,
                                         oversubscribed=True),
    "ref_initial_running_4k_x_200": dict(seed=105, n_tasks=4000, n_servants=200, n_envs=2,
                                         initial_running=True),
    # > 64 distinct compiler digests: (n, env_words) masks
    "ref_wide_envs_150_4k_x_300": dict(seed=106, n_tasks=4000, n_servants=300, n_envs=150,
                                       unknown_env_frac=0.01, self_frac=0.2),
}

# Full-size pools of BASELINE.json configs[2] / configs[3]: the first PREFIX requests through
# the verbatim reference (a sequential batch's prefix is the prefix batch, so this pins the
# first PREFIX placements of the full batch). The pools are regenerated from their seeds
# (yadcc_amd.synth.make_config); only the reference's answers and a checksum of the inputs
# are stored.
PREFIX = 50_000
PREFIX_SPECS = {"ref_cfg3_prefix_50k": "cfg3", "ref_cfg4_prefix_50k": "cfg4"}


def input_checksum(sv, tk, n):
    """Order-sensitive digest of the pool and the first n requests (guards against generator
    drift between the committed answers and the regenerated inputs)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    return h.hexdigest()

if __name__ == "__main__":
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, kw in SPECS.items():
        sv, tk = cases.random_case(**kw)
        d = R.RefDispatcher()
        d.load_servants(sv)
        idx, _, secs, _ = d.dispatch_batch(tk)
        d.close()
        granted = idx < R.IDX_ENV_NOT_FOUND
        run = sv["running_tasks"] + np.bincount(idx[granted], minlength=len(sv["version"])).astype(
            np.uint32)
        blob = {"sv_" + k: v for k, v in sv.items()}
        blob.update({"tk_" + k: v for k, v in tk.items()})
        blob["ref_servant_idx"] = idx
        blob["ref_running_after"] = run.astype(np.uint32)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **blob)
        print(name, "granted", int(granted.sum()), "of", len(idx), "ref %.3fs" % secs)
    from yadcc_amd import synth  # noqa: E402
    for name, cfg in PREFIX_SPECS.items():
        sv, tk = synth.make_config(cfg)
        head = {k: v[:PREFIX] for k, v in tk.items()}
        d = R.RefDispatcher()
        d.load_servants(sv)
        idx, _, secs, _ = d.dispatch_batch(head)
        d.close()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), config=cfg, prefix=PREFIX,
                            input_sha256=input_checksum(sv, tk, PREFIX), ref_servant_idx=idx)
        print(name, "granted", int((idx < R.IDX_ENV_NOT_FOUND).sum()), "of", len(idx),
              "ref %.1fs" % secs)
