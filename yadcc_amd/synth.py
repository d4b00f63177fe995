"""Synthetic (servant pool, pending task batch) snapshots.

Column layout == include/yadcc_dispatch.h. The distributions follow
SURVEY.md §8(d): capacity rules from the reference daemon
(yadcc/daemon/cloud/execution_engine.cc:132,153: 95 % of cores on dedicated
servants, 40 % on user machines), the 10 G low-memory threshold
(yadcc/scheduler/task_dispatcher.cc:35-38), daemon version 20
(yadcc/daemon/common_flags.cc:63). Seeds: 42 pool, 43 tasks, 44 stream.
"""
import numpy as np

PRIORITY_DEDICATED = 1  # api/scheduler.proto:43
PRIORITY_USER = 2       # api/scheduler.proto:47
GIB = 1 << 30


def make_servants(n, n_tasks_hint=None, n_envs=1, seed=42, p_set=(64, 96, 128, 192, 256),
                  oversubscribed=False, shared_ip_frac=0.0, disjoint_envs=False):
    rng = np.random.default_rng(seed)
    p_set = np.asarray(p_set, dtype=np.int64)
    if n_tasks_hint:
        # Scale the core counts so that the pool's usable slots are ~1.25 N (0.5 N when oversubscribed).
        mean_max = (0.3 * 0.95 + 0.7 * 0.40) * p_set.mean() * 0.97 * 0.8 * 0.95
        want = (0.5 if oversubscribed else 1.25) * n_tasks_hint / max(n, 1)
        scale = want / mean_max
        p_set = np.maximum(2, np.round(p_set * scale).astype(np.int64))
    nproc = rng.choice(p_set, size=n)
    dedicated = rng.random(n) < 0.3
    max_tasks = np.where(dedicated, (nproc * 95) // 100, (nproc * 40) // 100)
    max_tasks = np.where(rng.random(n) < 0.03, 0, max_tasks)
    load = (rng.random(n) * 1.25 * nproc).astype(np.int64)
    version = np.where(rng.random(n) < 0.05, 19, 20)
    total_mem = np.full(n, 256 * GIB, dtype=np.uint64)
    avail = np.where(rng.random(n) < 0.05, rng.integers(1 * GIB, 10 * GIB, n),
                     rng.integers(16 * GIB, 200 * GIB, n)).astype(np.uint64)
    if n_envs == 1:
        env_mask = np.ones(n, dtype=np.uint64)
    elif n_envs > 64:
        # More digests than one mask word holds: (n, env_words) masks, every servant advertises
        # 1-5 of the digests (the reference keeps an unbounded list per servant,
        # yadcc/scheduler/task_dispatcher.h:93-94).
        words = (n_envs + 63) // 64
        env_mask = np.zeros((n, words), dtype=np.uint64)
        for s in range(n):
            for j in rng.choice(n_envs, size=int(rng.integers(1, 6)), replace=False):
                env_mask[s, j // 64] |= np.uint64(1) << np.uint64(j % 64)
    elif disjoint_envs:
        env_mask = (np.uint64(1) << rng.integers(0, n_envs, n).astype(np.uint64))
    else:
        bits = rng.random((n, n_envs)) < 0.5
        empty = ~bits.any(axis=1)
        while empty.any():
            bits[empty] = rng.random((int(empty.sum()), n_envs)) < 0.5
            empty = ~bits.any(axis=1)
        env_mask = (bits.astype(np.uint64) << np.arange(n_envs, dtype=np.uint64)).sum(
            axis=1).astype(np.uint64)
    ip = (10 << 24) + 1 + np.arange(n, dtype=np.int64)  # 10.a.b.c, unique
    port = np.full(n, 8335, dtype=np.int64)
    if shared_ip_frac > 0 and n > 1:
        # Some servants share a host with an EARLIER one (different port).
        share = np.nonzero(rng.random(n) < shared_ip_frac)[0]
        share = share[share > 0]
        ip[share] = ip[(rng.random(len(share)) * share).astype(np.int64)]
        port[share] = 8336 + np.arange(len(share))
    return {
        "version": version.astype(np.uint32),
        "num_processors": nproc.astype(np.uint32),
        "current_load": load.astype(np.uint32),
        "max_tasks": max_tasks.astype(np.uint32),
        "running_tasks": np.zeros(n, dtype=np.uint32),
        "priority": np.where(dedicated, PRIORITY_DEDICATED, PRIORITY_USER).astype(np.uint32),
        "total_memory": total_mem,
        "memory_available": avail,
        "env_mask": env_mask,
        "ip": ip.astype(np.uint32),
        "port": port.astype(np.uint32),
    }


def make_tasks(n, servants, n_envs=1, seed=43, unknown_env_frac=0.0, self_frac=0.10,
               min_version_20_frac=0.5):
    rng = np.random.default_rng(seed)
    env = rng.integers(0, n_envs, n).astype(np.int64)
    if unknown_env_frac > 0:
        env = np.where(rng.random(n) < unknown_env_frac, 0xFFFF, env)
    minv = np.where(rng.random(n) < min_version_20_frac, 20, 0)
    rip = (172 << 24) + (16 << 16) + rng.integers(0, 1 << 20, n)  # 172.16.0.0/12
    s_n = len(servants["ip"])
    if self_frac > 0 and s_n:
        own = rng.random(n) < self_frac
        rip = np.where(own, servants["ip"][rng.integers(0, s_n, n)].astype(np.int64), rip)
    return {
        "env_id": env.astype(np.uint32),
        "min_version": minv.astype(np.uint32),
        "requestor_ip": rip.astype(np.uint32),
    }


CONFIGS = {
    # name: (n_tasks, n_servants, n_envs, unknown_env_frac)   — BASELINE.json `configs`
    "cfg1": (1_000, 64, 1, 0.0),
    "cfg2": (100_000, 2_000, 1, 0.0),
    "cfg3": (1_000_000, 8_000, 4, 0.001),
    "cfg4": (4_000_000, 16_000, 4, 0.0),
    "cfg5": (10_000, 2_000, 1, 0.0),  # per tick
}


def make_config(name, oversubscribed=False, n_tasks=None, n_servants=None, n_envs=None, **kw):
    """n_envs: another number of compiler digests than the configuration's own (> 64: every
    servant advertises its own handful of them — about one servant class per servant)."""
    n, s, e, unk = CONFIGS[name]
    n = n_tasks or n
    s = n_servants or s
    e = n_envs or e
    hint = n if name != "cfg5" else 100_000
    sv = make_servants(s, n_tasks_hint=hint, n_envs=e, oversubscribed=oversubscribed,
                       **{k: v for k, v in kw.items() if k in ("seed", "disjoint_envs",
                                                              "shared_ip_frac")})
    tk = make_tasks(n, sv, n_envs=e, unknown_env_frac=unk)
    return sv, tk


def placement_hash(servant_idx):
    """Order-sensitive 64-bit digest of a placement vector (FNV-1a over the u32s)."""
    a = np.ascontiguousarray(servant_idx, dtype=np.uint32)
    h = np.uint64(0xCBF29CE484222325)
    # Vectorised polynomial variant: sum(idx[i] * P^(n-i)) mod 2^64, order sensitive.
    with np.errstate(over="ignore"):
        p = np.uint64(0x100000001B3)
        powers = np.cumprod(np.full(len(a), p, dtype=np.uint64)[::-1])[::-1] if len(a) else a
        h = h + (a.astype(np.uint64) * powers).sum(dtype=np.uint64)
    return int(h)
