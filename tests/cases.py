"""Seeded snapshot families shared by the CPU (oracle/model) and GPU parity tests."""
import numpy as np

from yadcc_amd import synth

GIB = 1 << 30


def random_case(seed, n_tasks, n_servants, n_envs=1, **kw):
    sv_kw = {k: kw[k] for k in ("oversubscribed", "shared_ip_frac", "disjoint_envs") if k in kw}
    tk_kw = {k: kw[k] for k in ("unknown_env_frac", "self_frac", "min_version_20_frac") if k in kw}
    sv = synth.make_servants(n_servants, n_tasks_hint=n_tasks, n_envs=n_envs, seed=seed, **sv_kw)
    rng = np.random.default_rng(seed + 7)
    if kw.get("initial_running"):
        top = np.minimum(sv["max_tasks"], sv["num_processors"]).astype(np.int64)
        sv["running_tasks"] = (rng.random(n_servants) * (top + 3)).astype(np.uint32)
    if kw.get("no_memory_report"):
        sv["total_memory"][rng.random(n_servants) < 0.5] = 0  # total_memory == 0: not reported
    tk = synth.make_tasks(n_tasks, sv, n_envs=n_envs, seed=seed + 100, **tk_kw)
    return sv, tk


SMALL_CASES = [
    # name, kwargs
    ("cfg1_plumbing", dict(seed=1, n_tasks=1000, n_servants=64)),
    ("single_env_self30", dict(seed=2, n_tasks=4000, n_servants=150, self_frac=0.3)),
    ("four_envs", dict(seed=3, n_tasks=5000, n_servants=300, n_envs=4, unknown_env_frac=0.002,
                       self_frac=0.3)),
    ("four_envs_disjoint", dict(seed=4, n_tasks=5000, n_servants=300, n_envs=4,
                                disjoint_envs=True)),
    ("shared_ips", dict(seed=5, n_tasks=3000, n_servants=100, n_envs=4, shared_ip_frac=0.25,
                        self_frac=0.33)),
    ("oversubscribed", dict(seed=6, n_tasks=20000, n_servants=200, n_envs=4, oversubscribed=True,
                            unknown_env_frac=0.002)),
    ("initial_running", dict(seed=7, n_tasks=3000, n_servants=200, n_envs=2,
                             initial_running=True)),
    ("no_memory_report", dict(seed=8, n_tasks=3000, n_servants=120, no_memory_report=True)),
    ("tiny_pool_self_fallback", dict(seed=9, n_tasks=200, n_servants=5, self_frac=0.8)),
    ("one_servant", dict(seed=10, n_tasks=50, n_servants=1, self_frac=1.0)),
    ("many_envs", dict(seed=11, n_tasks=6000, n_servants=400, n_envs=6, unknown_env_frac=0.01)),
    # > 64 distinct digests: multi-word environment masks (no limit in the reference)
    ("wide_envs_150", dict(seed=12, n_tasks=4000, n_servants=300, n_envs=150,
                           unknown_env_frac=0.01, self_frac=0.2)),
]


def handmade_cases():
    """Edge cases the reference's own tests do not cover (last paragraph)."""
    out = []

    def sv(rows):
        cols = ("version", "num_processors", "current_load", "max_tasks", "running_tasks",
                "priority", "total_memory", "memory_available", "env_mask", "ip", "port")
        d = {c: [] for c in cols}
        for r in rows:
            base = dict(version=20, num_processors=16, current_load=0, max_tasks=8,
                        running_tasks=0, priority=2, total_memory=64 * GIB,
                        memory_available=32 * GIB, env_mask=1, ip=0x0A000001, port=8335)
            base.update(r)
            for c in cols:
                d[c].append(base[c])
        u64 = ("total_memory", "memory_available", "env_mask")
        return {c: np.array(v, dtype=np.uint64 if c in u64 else np.uint32) for c, v in d.items()}

    def tk(rows):
        d = {"env_id": [], "min_version": [], "requestor_ip": []}
        for r in rows:
            base = dict(env_id=0, min_version=0, requestor_ip=0xAC100001)
            base.update(r)
            for c in d:
                d[c].append(base[c])
        return {c: np.array(v, dtype=np.uint32) for c, v in d.items()}

    # empty batch / empty pool
    out.append(("empty_batch", sv([{}]), tk([])))
    out.append(("empty_pool", sv([]), tk([{}] * 3)))
    # max_tasks == 0 everywhere: env exists only on non-accepting servants => ENV_NOT_FOUND
    out.append(("all_not_accepting", sv([dict(max_tasks=0), dict(max_tasks=0, ip=0x0A000002)]),
                tk([{}] * 4)))
    # load >= nproc, load < running, running > capacity
    out.append(("capacity_corner", sv([
        dict(current_load=16, ip=0x0A000001), dict(current_load=40, ip=0x0A000002),
        dict(current_load=3, running_tasks=5, ip=0x0A000003),
        dict(max_tasks=4, running_tasks=9, ip=0x0A000004),
        dict(current_load=15, max_tasks=8, ip=0x0A000005)]), tk([{}] * 30)))
    # low memory / memory not reported
    out.append(("low_memory", sv([
        dict(memory_available=1 * GIB, ip=0x0A000001),
        dict(memory_available=1 * GIB, total_memory=0, ip=0x0A000002, max_tasks=3)]),
        tk([{}] * 6)))
    # min_version filtering incl. version compared as unsigned
    out.append(("min_version", sv([
        dict(version=19, ip=0x0A000001), dict(version=20, ip=0x0A000002, max_tasks=2),
        dict(version=0xFFFFFFFF, ip=0x0A000003, max_tasks=1)]),
        tk([dict(min_version=20)] * 4 + [dict(min_version=0)] * 3 + [dict(min_version=21)] * 2)))
    # dedicated tier: dedicated below 50 % of nproc beats an idle user servant
    out.append(("dedicated_tier", sv([
        dict(priority=2, ip=0x0A000001, max_tasks=16),
        dict(priority=1, ip=0x0A000002, max_tasks=15, current_load=2)]), tk([{}] * 31)))
    # requestor == only servant: self fallback
    out.append(("self_only", sv([dict(ip=0xAC100001)]), tk([{}] * 10)))
    # two servants on the requestor's host: only the first free one is `self`
    out.append(("two_on_host", sv([
        dict(ip=0xAC100001, port=1, max_tasks=2), dict(ip=0xAC100001, port=2, max_tasks=3),
        dict(ip=0x0A000009, max_tasks=2)]), tk([{}] * 9)))
    # burst from one servant host while it owns the globally best slots
    out.append(("burst_from_best", sv([
        dict(ip=0xAC100001, priority=1, max_tasks=15, num_processors=32),
        dict(ip=0x0A000002, max_tasks=6), dict(ip=0x0A000003, max_tasks=6)]),
        tk([{}] * 12 + [dict(requestor_ip=0xAC100009)] * 3 + [{}] * 14)))
    # heterogeneous digests with an unknown one in the middle
    out.append(("env_mix", sv([
        dict(env_mask=0b01, ip=0x0A000001, max_tasks=3), dict(env_mask=0b10, ip=0x0A000002, max_tasks=3),
        dict(env_mask=0b11, ip=0x0A000003, max_tasks=3)]),
        tk([dict(env_id=0), dict(env_id=1), dict(env_id=77), dict(env_id=1), dict(env_id=0)] * 3)))
    # huge capacities: forces the fp64 key path (capacity >= 2^21)
    out.append(("huge_capacity", sv([
        dict(num_processors=3_000_000, max_tasks=3_000_000, running_tasks=2_999_990, ip=0x0A000001),
        dict(num_processors=5_000_011, max_tasks=5_000_000, running_tasks=4_999_985,
             current_load=4_999_000, ip=0x0A000002, priority=1)]), tk([{}] * 30)))
    return out


def huge_capacity_pool(seed=31, n_servants=40, n_tasks=2500, n_envs=2):
    """Capacities of 2^21 .. 2^23 with all but a few dozen slots taken: the fp64-key path
    (task_dispatcher.cc:440-447 divides doubles; the integer key of the smaller pools is not
    exact here). Utilisations differ in the 6th .. 7th digit; dedicated servants with
    2 r < nproc and user servants, foreign load around `running` so that capacity moves with r."""
    rng = np.random.default_rng(seed)
    sv = synth.make_servants(n_servants, n_envs=n_envs, seed=seed)
    nproc = rng.integers(1 << 21, 1 << 23, n_servants)
    dedicated = sv["priority"] == synth.PRIORITY_DEDICATED
    max_tasks = np.where(dedicated, (nproc * 95) // 100, (nproc * 40) // 100)
    headroom = rng.integers(1, 90, n_servants)
    # dedicated servants: some start below the half-way mark (tier 0) and cross it inside the
    # batch (their max_tasks ends a few dozen slots above it), some start above
    low = dedicated & (rng.random(n_servants) < 0.5)
    max_tasks = np.where(low, nproc // 2 + rng.integers(1, 60, n_servants), max_tasks)
    run = np.where(low, nproc // 2 - headroom, max_tasks - headroom)
    sv["num_processors"] = nproc.astype(np.uint32)
    sv["max_tasks"] = max_tasks.astype(np.uint32)
    sv["running_tasks"] = run.astype(np.uint32)
    sv["current_load"] = np.maximum(0, run + rng.integers(-40, 40, n_servants)).astype(np.uint32)
    sv["memory_available"][:] = 64 * GIB
    tk = synth.make_tasks(n_tasks, sv, n_envs=n_envs, seed=seed + 100, self_frac=0.2)
    return sv, tk


# ---------------------------------------------------------------------------
# Full-size pools pinned to the VERBATIM reference: tests/golden/ref_cfg{3,4}_prefix_50k.npz hold
# the reference's placement of the first 50k requests of the cfg3 / cfg4 batches (generated by
# tests/golden/make_golden.py). A sequential batch's prefix is the prefix batch, so the first
# 50k answers of any full-size run must equal them.
# ---------------------------------------------------------------------------
def reference_prefix(cfg, sv, tk):
    """Returns the reference's servant indexes for the first requests of config `cfg`, after
    checking that (sv, tk) are the inputs the fixture was generated from."""
    import hashlib
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "ref_%s_prefix_50k.npz" % cfg))
    n = int(z["prefix"])
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    assert h.hexdigest() == str(z["input_sha256"]), "generator drift: regenerate tests/golden"
    return z["ref_servant_idx"]


def check_prefix_digests(cfg, sv, tk, servant_idx):
    """Compares the first placements of a full-size run with the verbatim reference's block digests
    (tests/golden/ref_<cfg>_prefix_digests.npz, generator tests/golden/make_prefix_digests.py:
    cfg3's first 400k requests — the dedicated-tier boundary and its chain included —, cfg4's
    first 200k). Returns the number of requests covered."""
    import hashlib
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "ref_%s_prefix_digests.npz" % cfg))
    n, block = int(z["prefix"]), int(z["block"])
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    assert h.hexdigest() == str(z["input_sha256"]), "generator drift: regenerate tests/golden"
    for b in range(n // block):
        part = servant_idx[b * block:(b + 1) * block]
        assert synth.placement_hash(part) == int(z["digest"][b]), "requests %d..%d differ from the reference" % (
            b * block, (b + 1) * block)
        assert int((part < 0xFFFFFFFE).sum()) == int(z["granted"][b])
    return n
