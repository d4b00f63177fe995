"""Parity tests proper: the HIP path (through the C-ABI, yadcc_amd/libydc.so) against the
oracle on the same seeded snapshots. Bit-exact placement and running_tasks; the
chosen-servant utilisation is the same IEEE double division on both sides, so it is
compared exactly too (north_star tolerance: 1e-6)."""
import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from yadcc_amd import binding, pack, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = binding.Context(device=0)
    yield c
    c.close()


def run_gpu(ctx, sv, tk, **kw):
    ctx.upload_servants(pack.to_abi_columns(sv))
    return ctx.dispatch(tk, **kw)


def check(ctx, sv, tk, method="sorted"):
    want, wutil, wrun = O.dispatch(sv, tk, method)
    got, gutil, grun = run_gpu(ctx, sv, tk)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at task %d: gpu %d oracle %d (%d total) stats=%s" % (
        bad[0], got[bad[0]], want[bad[0]], bad.size, ctx.stats())
    assert np.array_equal(grun, wrun)
    assert np.allclose(gutil, wutil, rtol=0, atol=1e-6) and np.array_equal(gutil, wutil)
    st = ctx.stats()
    assert st["granted"] == int((want < O.IDX_ENV_NOT_FOUND).sum())
    assert st["timeouts"] == int((want == O.IDX_TIMEOUT).sum())
    assert st["env_not_found"] == int((want == O.IDX_ENV_NOT_FOUND).sum())
    return st


@pytest.mark.parametrize("name,kw", cases.SMALL_CASES, ids=[c[0] for c in cases.SMALL_CASES])
def test_small_cases(ctx, name, kw):
    sv, tk = cases.random_case(**kw)
    check(ctx, sv, tk, "scan")  # literal restatement of the reference


@pytest.mark.parametrize("name,sv,tk", cases.handmade_cases(),
                         ids=[c[0] for c in cases.handmade_cases()])
def test_handmade_cases(ctx, name, sv, tk):
    check(ctx, sv, tk, "scan")


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_fp64_key_path_against_the_verbatim_reference(ctx, seed):
    """Capacities of 2^21 .. 2^23 (key_bits == 64): the HIP path against the reference class itself
    (oracle/_ref travels to the GPU box; its running_tasks are preset through the FRIEND_TEST
    door of oracle/ref_driver.cc), and against the literal restatement."""
    from oracle import refbind as R
    sv, tk = cases.huge_capacity_pool(seed=seed)
    st = check(ctx, sv, tk, "scan")
    assert st["key_bits"] == 64
    if not R.available():
        pytest.skip("oracle/_ref is not built on this box")
    d = R.RefDispatcher()
    d.load_servants(sv)
    ref_idx, _, _, _ = d.dispatch_batch(tk)
    d.close()
    got, _, _ = run_gpu(ctx, sv, tk)
    assert np.array_equal(got, ref_idx)


def test_handmade_huge_capacity_against_the_verbatim_reference(ctx):
    from oracle import refbind as R
    if not R.available():
        pytest.skip("oracle/_ref is not built on this box")
    name, sv, tk = [c for c in cases.handmade_cases() if c[0] == "huge_capacity"][0]
    d = R.RefDispatcher()
    d.load_servants(sv)
    ref_idx, _, _, _ = d.dispatch_batch(tk)
    d.close()
    got, _, _ = run_gpu(ctx, sv, tk)
    assert ctx.stats()["key_bits"] == 64 and np.array_equal(got, ref_idx)


@pytest.mark.parametrize("shape", ["cfg2", "ragged", "oversubscribed"])
def test_dispatch_device_caller_owned_outputs(ctx, shape):
    """ydc_dispatch_device — the entry point bench.py times — called directly: request columns
    and all three outputs in caller-owned device buffers (poisoned first), against the oracle.
    cfg2 = BASELINE.json configs[1]; a ragged size (not a multiple of a wave, a chunk or a sort
    tile) with self requests; a pool half the size of the batch (Timeout tail)."""
    DA = binding.DeviceArray
    if shape == "cfg2":
        sv, tk = synth.make_config("cfg2")
    elif shape == "ragged":
        sv, tk = cases.random_case(seed=91, n_tasks=33_337, n_servants=777, n_envs=3, self_frac=0.2,
                                   unknown_env_frac=0.003)
    else:
        sv, tk = cases.random_case(seed=92, n_tasks=50_001, n_servants=400, n_envs=2,
                                   oversubscribed=True)
    n, S = len(tk["env_id"]), len(sv["version"])
    ctx.upload_servants(pack.to_abi_columns(sv))
    cols = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    d_idx = DA.from_numpy(np.full(n, 0xDEADBEEF, np.uint32))
    d_util = DA.from_numpy(np.full(n, -7.0, np.float64))
    d_run = DA.from_numpy(np.full(S, 0xDEADBEEF, np.uint32))
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    for rep in range(2):  # the second call reuses the workspace and the same output buffers
        ctx.dispatch_device(cols[0], cols[1], cols[2], d_idx, d_util, d_run)
        got = d_idx.numpy()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (rep, bad[:5], got[bad[:5]], want[bad[:5]], ctx.stats())
        assert np.array_equal(d_util.numpy(), wutil) and np.array_equal(d_run.numpy(), wrun)
    # outputs are optional: placement only, then the resident column is still untouched (no COMMIT)
    ctx.dispatch_device(cols[0], cols[1], cols[2], d_idx)
    assert np.array_equal(d_idx.numpy(), want)
    assert np.array_equal(ctx.get_running(), np.asarray(sv["running_tasks"], np.uint32))
    ctx.dispatch_device(cols[0], cols[1], cols[2], d_idx, None, None, commit=True)
    assert np.array_equal(ctx.get_running(), wrun)
    if shape == "oversubscribed":
        assert (want == O.IDX_TIMEOUT).sum() > 1000


@pytest.mark.parametrize("host_in", ["map", "copy"])
def test_dispatch_zero_copy_pinned_buffers(host_in, monkeypatch):
    """ydc_dispatch with page-locked caller buffers (ydc_host_alloc / ydc_host_register): the
    kernels read the request columns and write idx / utilisation / running_tasks through the
    buffers' device addresses (YDC_HOST_IN=map), or DMA straight from them (copy) — no staging.
    Same answers as with pageable buffers and as the oracle; mixed pinned / pageable arguments
    take the staged path."""
    monkeypatch.setenv("YDC_HOST_IN", host_in)
    c = binding.Context(device=0)
    sv, tk = cases.random_case(seed=93, n_tasks=70_001, n_servants=1500, n_envs=4, self_frac=0.2,
                               unknown_env_frac=0.002)
    n, S = len(tk["env_id"]), len(sv["version"])
    c.upload_servants(pack.to_abi_columns(sv))
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    # (1) everything allocated page-locked
    pin = {k: binding.pinned_empty(n, np.uint32) for k in tk}
    for k in tk:
        pin[k][:] = tk[k]
    out = binding.pinned_empty(n, np.uint32)
    out[:] = 0xDEADBEEF
    for rep in range(2):
        got, util, run = c.dispatch(pin, out_idx=out)
        assert got is out and np.array_equal(out, want), (rep, c.stats())
        assert np.array_equal(util, wutil) and np.array_equal(run, wrun)
    # (2) the caller's own arrays registered in place; results into a registered array too
    reg = {k: np.ascontiguousarray(tk[k], dtype=np.uint32).copy() for k in tk}
    out2 = np.full(n, 0xDEADBEEF, np.uint32)
    for a in list(reg.values()) + [out2]:
        binding.host_register(a)
    try:
        c.dispatch(reg, want_util=False, want_running=False, out_idx=out2)
        assert np.array_equal(out2, want)
        # a batch that is a prefix of the registered columns (pointer inside a registered range)
        half = {k: v[:n // 2] for k, v in reg.items()}
        w2, _, _ = O.dispatch(sv, {k: tk[k][:n // 2] for k in tk}, "sorted")
        c.dispatch(half, want_util=False, want_running=False, out_idx=out2[:n // 2])
        assert np.array_equal(out2[:n // 2], w2)
        # (3) mixed: pinned columns, pageable result array -> staged results, same answer
        got3, _, _ = c.dispatch(reg, want_util=False, want_running=False)
        assert np.array_equal(got3, want)
    finally:
        for a in list(reg.values()) + [out2]:
            binding.host_unregister(a)
    # (4) after unregistering the very same arrays go through the staging path again
    got4, _, _ = c.dispatch(reg, want_util=False, want_running=False, out_idx=out2)
    assert np.array_equal(got4, want)
    c.close()


def test_pipelined_batches_commit_in_order(monkeypatch):
    """ydc_dispatch_device_async / ydc_dispatch_wait: a batch is enqueued while the previous one is
    still running; every batch COMMITs, so batch k + 1 must see exactly the registry batch k left.
    Six batches of one sequence == the oracle on the concatenation. Then the same with matching
    passes that cannot converge in the pipeline (hand-off patience 0: most waves leave their
    chunk to a later pass than the pipeline enqueues): the batches are replayed in order by
    ydc_dispatch_wait — same placement, same final registry."""
    DA = binding.DeviceArray
    sv, tk = cases.random_case(seed=95, n_tasks=120_000, n_servants=1800, n_envs=3, self_frac=0.15,
                               unknown_env_frac=0.002)
    want, _, wrun = O.dispatch(sv, tk, "sorted")
    n, S, nb = len(tk["env_id"]), len(sv["version"]), 6
    per = n // nb
    for tries, swap in ((None, "1"), ("0", "1"), (None, "0")):
        if tries is not None:
            monkeypatch.setenv("YDC_HAND_TRIES", tries)
        else:
            monkeypatch.delenv("YDC_HAND_TRIES", raising=False)
        # COMMIT by swapping the two running_tasks columns, and the copying COMMIT with the outcome
        # fetched by a blit instead of k_finalize's store (the round-4 path)
        monkeypatch.setenv("YDC_COMMIT_SWAP", swap)
        monkeypatch.setenv("YDC_OUTCOME_STORE", swap)
        c = binding.Context(device=0)
        c.upload_servants(pack.to_abi_columns(sv))
        cols = [[DA.from_numpy(tk[k][b * per:(b + 1) * per]) for k in ("env_id", "min_version", "requestor_ip")]
                for b in range(nb)]
        outs = [DA.from_numpy(np.full(per, 0xDEADBEEF, np.uint32)) for _ in range(nb)]
        runs = [DA(S, np.uint32) for _ in range(nb)]
        c.dispatch_device_async(*cols[0], outs[0], None, runs[0], commit=True)
        for b in range(1, nb):
            c.dispatch_device_async(*cols[b], outs[b], None, runs[b], commit=True)
            c.dispatch_wait()
        c.dispatch_wait()
        got = np.concatenate([o.numpy() for o in outs])
        bad = np.nonzero(got != want[:per * nb])[0]
        assert bad.size == 0, (tries, swap, bad[:5], got[bad[:5]], want[bad[:5]])
        wrun_b = O.dispatch(sv, {k: v[:per * nb] for k, v in tk.items()}, "sorted")[2]
        assert np.array_equal(runs[-1].numpy(), wrun_b) and np.array_equal(c.get_running(), wrun_b)
        with pytest.raises(binding.YdcError):
            c.dispatch_wait()  # nothing outstanding
        c.close()


def test_golden_load_balance(ctx):
    """task_dispatcher_test.cc:216-298 through the GPU path, one request per batch with the
    chosen servant re-heartbeated at load + 1, like the reference test."""
    from tests.test_oracle_golden import LB, LB_EXPECT, _lb_columns
    loads = [x[3] for x in LB]
    running = [0] * len(LB)
    one = {"env_id": np.zeros(1, np.uint32), "min_version": np.full(1, 8, np.uint32),
           "requestor_ip": np.array([0x7F000003], np.uint32)}
    sv0 = {k: v[:1] for k, v in _lb_columns(loads, running).items()}
    idx, _, _ = run_gpu(ctx, sv0, one)
    assert idx[0] == binding.IDX_TIMEOUT
    got = []
    for _ in LB_EXPECT:
        idx, _, run = run_gpu(ctx, _lb_columns(loads, running), one)
        got.append(int(idx[0]))
        running = run.tolist()
        loads[got[-1]] += 1
    assert got == LB_EXPECT


def test_golden_fixture_file(ctx):
    """Committed golden vectors (tests/golden/*.npz) generated from the verbatim reference."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    files = [f for f in files if "_prefix_" not in f and "_stream_" not in f]  # (those: test_cfg3_full_size, cfg4, streaming)
    assert len(files) >= 6, "golden fixtures missing"
    for f in files:
        z = np.load(f)
        sv = {k[3:]: z[k] for k in z.files if k.startswith("sv_")}
        tk = {k[3:]: z[k] for k in z.files if k.startswith("tk_")}
        got, _, grun = run_gpu(ctx, sv, tk)
        assert np.array_equal(got, z["ref_servant_idx"]), f
        assert np.array_equal(grun, z["ref_running_after"]), f


def test_cfg2_full_size(ctx):
    """BASELINE.json configs[1]: 100k tasks x 2k servants, single compiler env."""
    sv, tk = synth.make_config("cfg2")
    st = check(ctx, sv, tk)
    assert st["n_tasks"] == 100_000 and st["n_servants"] == 2000


@pytest.mark.parametrize("frac", [0.05, 0.5])
def test_cfg2_hosts_with_several_servants(ctx, frac):
    """configs[1]'s shape with servants that share hosts: the requestor avoids the FIRST free
    eligible servant on its host (task_dispatcher.cc:372-379), resolved at replay time in the
    chunk-parallel path — against the literal restatement of the reference."""
    sv, tk = synth.make_config("cfg2", shared_ip_frac=frac, n_tasks=40_000, n_servants=800)
    tk = synth.make_tasks(40_000, sv, self_frac=0.4)
    st = check(ctx, sv, tk, "scan")
    assert st["n_chunks"] > 100  # not one sequential chunk


def test_tiny_pool_with_huge_servants_and_own_host_traffic(ctx):
    """Five servants with tens of thousands of slots each, one per class, a tenth of the requests
    from their own hosts (every start state has holes: about one pass per chunk — exact, and
    quick since a one-servant class is known to hold nothing but own slots)."""
    sv, tk = cases.random_case(seed=2009, n_tasks=150_000, n_servants=5, n_envs=3, self_frac=0.1,
                               unknown_env_frac=0.01, initial_running=True)
    st = check(ctx, sv, tk)
    assert st["n_chunks"] > 1000


def test_cfg2_oversubscribed(ctx):
    sv, tk = synth.make_config("cfg2", oversubscribed=True)
    st = check(ctx, sv, tk)
    assert st["timeouts"] > 10_000


def test_cfg3_full_size(ctx):
    """BASELINE.json configs[2]: 1M tasks x 8k servants, 4 overlapping digests + env filtering."""
    sv, tk = synth.make_config("cfg3")
    st = check(ctx, sv, tk)
    assert st["n_classes"] >= 20 and st["env_not_found"] > 0
    # ... and pinned to the VERBATIM reference on the first 50k requests of this very batch.
    ref = cases.reference_prefix("cfg3", sv, tk)
    got, _, _ = ctx.dispatch(tk, want_util=False, want_running=False)
    assert np.array_equal(got[:len(ref)], ref)
    # ... and, block by block, on its first 400k requests: the dedicated-tier boundary (request ~337k)
    # and the chain of ~2000 requests the matching passes follow behind it.
    assert cases.check_prefix_digests("cfg3", sv, tk, got) == 400_000


def test_cfg4_full_size_one_gpu(ctx):
    """BASELINE.json configs[3]'s batch (4M requests x 16k servants) on ONE GPU: against the
    slot-order oracle, and its first 50k placements against the verbatim reference."""
    sv, tk = synth.make_config("cfg4")
    check(ctx, sv, tk)
    ref = cases.reference_prefix("cfg4", sv, tk)
    got, _, _ = ctx.dispatch(tk, want_util=False, want_running=False)
    assert np.array_equal(got[:len(ref)], ref)
    assert cases.check_prefix_digests("cfg4", sv, tk, got) == 200_000


def test_cfg3_disjoint_envs(ctx):
    """Disjoint environment partitions: independent parts of the registry, level guesses per
    part (the global level needed about one pass per chunk here)."""
    sv, tk = synth.make_config("cfg3", disjoint_envs=True, n_tasks=300_000)
    st = check(ctx, sv, tk)
    assert st["rounds"] <= 8, st


def test_properties_at_full_size(ctx):
    """Size-independent properties on cfg3: conservation, capacity, eligibility, and the
    monotonicity per (env, min_version) type of the utilisation keys."""
    sv, tk = synth.make_config("cfg3")
    got, util, run = run_gpu(ctx, sv, tk)
    granted = got < binding.IDX_ENV_NOT_FOUND
    # conservation: running_after - running_before == histogram of the placement
    hist = np.bincount(got[granted], minlength=len(sv["version"]))
    assert np.array_equal(run - sv["running_tasks"], hist)
    # never above min(max_tasks, nproc); never on low-memory / overloaded / max_tasks == 0
    top = np.minimum(sv["max_tasks"], sv["num_processors"])
    assert (run <= np.maximum(top, sv["running_tasks"])).all()
    flags = pack.servant_flags(sv)
    dead = ((flags & 2) != 0) | (sv["current_load"] >= sv["num_processors"]) | (sv["max_tasks"] == 0)
    assert hist[dead].sum() == 0
    # eligibility of every grant
    s = got[granted]
    env = tk["env_id"][granted].astype(np.uint64)
    assert ((sv["env_mask"][s] >> env) & np.uint64(1)).all()
    assert (sv["version"][s] >= tk["min_version"][granted]).all()
    # ENV_NOT_FOUND exactly for the unknown digest
    assert np.array_equal(got == binding.IDX_ENV_NOT_FOUND, tk["env_id"] >= 64)


def test_commit_then_second_batch(ctx):
    """COMMIT keeps the grants in the resident running_tasks (task_dispatcher.cc:123): two
    committed half batches == one full batch."""
    sv, tk = cases.random_case(seed=31, n_tasks=20_000, n_servants=500, n_envs=3, self_frac=0.2)
    want, _, wrun = O.dispatch(sv, tk, "sorted")
    ctx.upload_servants(pack.to_abi_columns(sv))
    half = {k: v[:10_000] for k, v in tk.items()}
    rest = {k: v[10_000:] for k, v in tk.items()}
    a, _, _ = ctx.dispatch(half, commit=True)
    b, _, run = ctx.dispatch(rest, commit=True)
    assert np.array_equal(np.concatenate([a, b]), want)
    assert np.array_equal(run, wrun) and np.array_equal(ctx.get_running(), wrun)


def test_heartbeat_update_and_release(ctx):
    """ydc_update_servants == KeepServantAlive of a known servant (personality replaced,
    running_tasks kept, task_dispatcher.cc:195-201); ydc_release_slots == FreeTask's
    --running_tasks (:181)."""
    sv, tk = cases.random_case(seed=32, n_tasks=4000, n_servants=120, n_envs=2)
    ctx.upload_servants(pack.to_abi_columns(sv))
    a, _, run1 = ctx.dispatch(tk, commit=True)
    # free every second grant, re-heartbeat 30 servants with a new load
    granted = a[a < binding.IDX_ENV_NOT_FOUND]
    freed = granted[::2]
    ctx.release_slots(freed)
    rng = np.random.default_rng(5)
    who = rng.choice(len(sv["version"]), 30, replace=False)
    sv2 = {k: v.copy() for k, v in sv.items()}
    sv2["current_load"][who] = (rng.random(30) * sv2["num_processors"][who]).astype(np.uint32)
    cols = pack.to_abi_columns(sv2)
    rows = [{k: cols[k][s] for k in ("version", "num_processors", "current_load", "max_tasks",
                                     "flags", "ip_id", "env_mask")} for s in who]
    ctx.update_servants(who, rows)
    sv2["running_tasks"] = (run1 - np.bincount(freed, minlength=len(run1))).astype(np.uint32)
    assert np.array_equal(ctx.get_running(), sv2["running_tasks"])
    _, tk2 = cases.random_case(seed=33, n_tasks=3000, n_servants=120, n_envs=2)
    want, _, wrun = O.dispatch(sv2, tk2, "sorted")
    got, _, grun = ctx.dispatch(tk2)
    assert np.array_equal(got, want) and np.array_equal(grun, wrun)


@pytest.mark.parametrize("chunk", [64, 256, 4096])
def test_chunk_size_does_not_change_results(chunk, monkeypatch):
    monkeypatch.setenv("YDC_CHUNK_SIZE", str(chunk))
    c = binding.Context(device=0)
    try:
        sv, tk = cases.random_case(seed=34, n_tasks=60_000, n_servants=900, n_envs=4,
                                   unknown_env_frac=0.001, self_frac=0.2)
        st = check(c, sv, tk)
        assert st["n_chunks"] == -(-60_000 // chunk)
    finally:
        c.close()


def test_many_classes_paths(ctx):
    """> 64 classes (two per lane) and > 256 classes (thread-per-chunk kernel)."""
    sv, tk = cases.random_case(seed=35, n_tasks=20_000, n_servants=1500, n_envs=7,
                               unknown_env_frac=0.002)
    st = check(ctx, sv, tk)
    assert 64 < st["n_classes"] <= 256
    sv, tk = cases.random_case(seed=36, n_tasks=20_000, n_servants=3000, n_envs=10)
    sv["version"] = (20 + np.arange(3000) % 3).astype(np.uint32)
    st = check(ctx, sv, tk)
    assert st["n_classes"] > 256


@pytest.mark.parametrize("wide", [1, 5, 4, 2, 3, 0])
def test_more_than_256_classes(wide, monkeypatch):
    """Pools whose machines advertise individual compiler sets (the reference has no limit on
    (environment set, version) combinations, task_dispatcher.h:93-94, .cc:316-344): 150 digests,
    about one class per servant. wide=1: one wave per chunk with the class states in LDS
    (k_sim_wide), a request's classes read from its (digest, version threshold) row — registries
    with such rows are walked 64 requests at a time from the first request on (k_walk_groups: head
    rank and class id in one word; wide=5: in two arrays, as registries with more slots need them);
    wide=4: rounds of speculation and then the lone walker instead; wide=2: the walk with prefetch waves; wide=3: mask
    scan instead of the rows; wide=0: the thread-per-chunk kernel. Plain, with traffic from the servants'
    own hosts on shared hosts (holes, `self` resolved at replay time), and oversubscribed."""
    monkeypatch.setenv("YDC_WIDE", "1" if wide else "0")
    if wide in (2, 4):
        monkeypatch.setenv("YDC_GROUP_WALK", "0")  # rounds, then one request at a time
    if wide == 5:
        monkeypatch.setenv("YDC_WALK_PACKED", "0")
    if wide == 2:
        monkeypatch.setenv("YDC_WALK_PREFETCH", "1")  # the walk with prefetch waves
    if wide == 3:
        monkeypatch.setenv("YDC_WIDE_LISTS", "0")  # mask scan instead of eligible-class lists
    c = binding.Context(device=0)
    try:
        n = 30_000 if wide else 6_000
        sv, tk = cases.random_case(seed=37, n_tasks=n, n_servants=1200, n_envs=150,
                                   unknown_env_frac=0.002, self_frac=0.1)
        st = check(c, sv, tk)
        assert 256 < st["n_classes"] <= 4096
        sv, tk = cases.random_case(seed=38, n_tasks=n // 3, n_servants=700, n_envs=150,
                                   shared_ip_frac=0.3, self_frac=0.5)
        st = check(c, sv, tk, "scan")
        assert st["n_classes"] > 256
        sv, tk = cases.random_case(seed=39, n_tasks=n, n_servants=900, n_envs=100,
                                   oversubscribed=True, self_frac=0.2)
        st = check(c, sv, tk)
        assert st["n_classes"] > 256 and st["timeouts"] > 100
    finally:
        c.close()


@pytest.mark.parametrize("seed", range(300, 312))
def test_group_walk_random_sparse_pools(seed, monkeypatch):
    """k_walk_groups (sparse eligibility: 40 .. 200 digests, about one class per servant) on random
    shapes: ragged batch sizes, none / heavy traffic from the servants' own hosts, shared hosts,
    oversubscription (Timeout tails), unknown digests, initial running_tasks, and a batch
    committed in two halves — placement, utilisation and running_tasks against the literal
    restatement of the reference."""
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, n_tasks=int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000])),
              n_servants=int(rng.choice([300, 700, 1500])), n_envs=int(rng.choice([40, 90, 150, 200])),
              self_frac=float(rng.choice([0.0, 0.1, 0.6])), unknown_env_frac=float(rng.choice([0.0, 0.01])),
              min_version_20_frac=float(rng.choice([0.0, 0.5, 1.0])))
    if rng.random() < 0.3:
        kw["oversubscribed"] = True
    if rng.random() < 0.3:
        kw["shared_ip_frac"] = 0.25
    if rng.random() < 0.3:
        kw["initial_running"] = True
    if seed % 4 == 1:
        monkeypatch.setenv("YDC_WALK_PACKED", "0")  # (head ranks and class ids in two arrays)
    if seed % 3 == 2:
        # the fetch of the entry after next in plain variables instead of parked in a254 / a255
        # (wide_kernel.h: plain_park) — the variant that rests on nothing the compiler does not know
        monkeypatch.setenv("YDC_WALK_PARK", "0")
    sv, tk = cases.random_case(**kw)
    c = binding.Context(device=0)
    try:
        want, wutil, wrun = O.dispatch(sv, tk, "scan" if len(tk["env_id"]) <= 5000 else "sorted")
        c.upload_servants(pack.to_abi_columns(sv))
        n = len(tk["env_id"])
        if seed % 3 == 0 and n > 1:
            cut = int(rng.integers(1, n))
            a, ua, _ = c.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
            b, ub, grun = c.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
            got, gutil = np.concatenate([a, b]), np.concatenate([ua, ub])
        else:
            got, gutil, grun = c.dispatch(tk)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (kw, bad[:5], got[bad[:5]], want[bad[:5]], c.stats())
        assert np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
    finally:
        c.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_ragged_batch_sizes(ctx, n):
    """Batches that are not a multiple of the 64-request block / of the chunk size."""
    sv, tk = cases.random_case(seed=40 + n, n_tasks=n, n_servants=30, n_envs=3, self_frac=0.3,
                               unknown_env_frac=0.05)
    check(ctx, sv, tk, "scan")


def test_single_class_no_self(ctx):
    """One class and nobody on a servant host: the merge degenerates to rank == request index."""
    sv, tk = cases.random_case(seed=51, n_tasks=30_000, n_servants=400, self_frac=0.0,
                               min_version_20_frac=0.0)
    sv["version"][:] = 20
    st = check(ctx, sv, tk)
    assert st["n_classes"] == 1 and st["rounds"] <= 2


@pytest.mark.parametrize("fused", [0, 1])
def test_class_partition_fused_into_last_sort_pass(fused, monkeypatch):
    """<= 8 classes and room in the last key digit: the class partition rides on the last key
    pass (k_radix_scatter_classed); YDC_FUSED_CLASS=0 keeps the separate pass. Same results."""
    monkeypatch.setenv("YDC_FUSED_CLASS", str(fused))
    monkeypatch.setenv("YDC_BINSORT", "0")  # (the radix sort is what is under test here)
    c = binding.Context(device=0)
    try:
        for seed, envs in ((61, 2), (62, 3), (63, 1)):
            sv, tk = cases.random_case(seed=seed, n_tasks=40_000, n_servants=700, n_envs=envs,
                                       self_frac=0.2, unknown_env_frac=0.001)
            st = check(c, sv, tk)
            assert 2 <= st["n_classes"] <= 16
        # fp64 keys (capacity >= 2^21): 64-bit sort keys in, 32-bit ranks out
        sv, tk = cases.random_case(seed=64, n_tasks=5_000, n_servants=50, n_envs=2)
        sv["num_processors"][:5] = 3_000_000
        sv["max_tasks"][:5] = 3_000_000
        sv["running_tasks"][:5] = 3_000_000 - 40
        st = check(c, sv, tk)
        assert st["key_bits"] == 64
    finally:
        c.close()


def test_hand_off_that_never_arrives(monkeypatch):
    """The launch of passes 0 + 1 waits for its predecessor's end state a bounded number of polls
    (HIP promises nothing about dispatch order). With no patience at all (YDC_HAND_TRIES=0) most
    waves give up, say "not final" and leave their chunk to the next launch: more passes, same
    placement."""
    monkeypatch.setenv("YDC_HAND_TRIES", "0")
    c = binding.Context(device=0)
    try:
        for seed, envs, n in ((74, 1, 30_000), (75, 3, 60_000)):
            sv, tk = cases.random_case(seed=seed, n_tasks=n, n_servants=700, n_envs=envs,
                                       self_frac=0.2, unknown_env_frac=0.001)
            st = check(c, sv, tk)
            assert st["rounds"] >= 2
        sv, tk = synth.make_config("cfg2")
        check(c, sv, tk)
    finally:
        c.close()


@pytest.mark.parametrize("cp_every", ["1", "2", "4", "1024"])
@pytest.mark.parametrize("patient", [True, False])
def test_checkpoint_spacing(cp_every, patient, monkeypatch):
    """Checkpoints before every block of 64 requests of a chunk (round 5), every second, every
    fourth (the default) or only the chunk's first (match_kernel.h: MatchBuffers::cp_every) — with
    chunks of 512 requests, and with waves that do not wait for their predecessor's hand-off
    (YDC_HAND_TRIES=0), so that most chunks are replayed by later launches, which is where the
    early stops at checkpoints decide how much of a chunk is replayed: same placement every time."""
    monkeypatch.setenv("YDC_CP_EVERY", cp_every)
    monkeypatch.setenv("YDC_CHUNK_SIZE", "512")
    if not patient:
        monkeypatch.setenv("YDC_HAND_TRIES", "0")
    c = binding.Context(device=0)
    try:
        for seed, envs, n in ((171, 1, 30_000), (172, 4, 80_000), (173, 6, 60_000)):
            sv, tk = cases.random_case(seed=seed, n_tasks=n, n_servants=900, n_envs=envs, self_frac=0.3,
                                       unknown_env_frac=0.001, oversubscribed=seed == 173)
            st = check(c, sv, tk)
            assert patient or st["rounds"] >= 2
    finally:
        c.close()


def test_switchable_fast_paths_off(monkeypatch):
    """The A/B switches select older, slower forms of the same steps (separate class pass,
    class gather, k_guess_init, one request per loop iteration, the radix sort, a launch of
    its own for pass 1): same results."""
    for k in ("YDC_PACKED_CLASS", "YDC_FUSED_CLASS", "YDC_OWN_GUESS", "YDC_PAIR", "YDC_BINSORT",
              "YDC_FUSE_PASSES"):
        monkeypatch.setenv(k, "0")
    c = binding.Context(device=0)
    try:
        for seed, envs, n in ((71, 1, 20_000), (72, 2, 30_000), (73, 4, 60_000)):
            sv, tk = cases.random_case(seed=seed, n_tasks=n, n_servants=800, n_envs=envs,
                                       self_frac=0.2, unknown_env_frac=0.001)
            check(c, sv, tk)
    finally:
        c.close()


@pytest.mark.parametrize("n_envs", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("rings", ["default", "1024", "256"])
def test_matching_loop_variants(n_envs, rings, monkeypatch):
    """The hand-scheduled loop of the matching kernel in all its shapes (match_kernel.h): one body
    per DPP depth (the digests make 2 .. 60 classes), single steps and pairs — with requests
    nobody can serve any more inside the pairs (an oversubscribed pool: half of the requests time
    out), requests for unknown digests (empty class masks) and the requestors' own servants —,
    chunks long enough for the 4-waves-per-SIMD build, and rings of 32 entries (watched by the loop
    itself), of 8 (the caller's bound) and whatever the plan picks; two committed batches."""
    monkeypatch.setenv("YDC_CHUNK_SIZE", "512")
    if rings != "default":
        monkeypatch.setenv("YDC_RING_TOTAL", rings)
    c = binding.Context(device=0)
    try:
        sv, tk = cases.random_case(seed=90 + n_envs, n_tasks=40_000, n_servants=700, n_envs=n_envs,
                                   oversubscribed=True, unknown_env_frac=0.01, self_frac=0.25)
        want, wutil, wrun = O.dispatch(sv, tk, "sorted")
        assert (want == O.IDX_TIMEOUT).sum() > 1000 and (want == O.IDX_ENV_NOT_FOUND).sum() > 100
        c.upload_servants(pack.to_abi_columns(sv))
        cut = 23_456
        a, ua, _ = c.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
        sa = c.stats()
        b, ub, run = c.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
        got = np.concatenate([a, b])
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "first mismatch at task %d: gpu %d oracle %d (%d total) stats=%s" % (
            bad[0], got[bad[0]], want[bad[0]], bad.size, sa)
        assert np.array_equal(np.concatenate([ua, ub]), wutil)
        assert np.array_equal(run, wrun) and np.array_equal(c.get_running(), wrun)
        print("n_envs", n_envs, "classes", sa["n_classes"], "rings", rings, "rounds", sa["rounds"])
    finally:
        c.close()


@pytest.mark.parametrize("lead", [None, "256", "off", "auto"])
def test_walk_of_the_dedicated_tiers_end(lead, monkeypatch):
    """zone_guess.h: workgroup 0 of the first matching launch walks the stretch where the dedicated
    tier runs out and the chunks there start from its cursors instead of their level guesses
    (cfg3's registry, a batch that reaches past the tier's end). Same placement as the oracle with
    the walk, without it, and with a walk that starts too late to be of use (lead 256: the served
    chunks replay again like any wrongly started chunk — and the next batch's walk starts earlier).
    "auto" (the default): no walk until a few batches have shown a chain behind the first launch,
    then the walk as long as batches cost less on the device with it."""
    monkeypatch.setenv("YDC_ZONE_GUESS", {"off": "0", "auto": "1"}.get(lead, "2"))
    if lead == "256":
        monkeypatch.setenv("YDC_ZONE_LEAD", lead)
    sv, tk = synth.make_config("cfg3")
    n = 480_000
    tk = {k: v[:n] for k, v in tk.items()}
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    c = binding.Context(device=0)
    try:
        c.upload_servants(pack.to_abi_columns(sv))
        seen = []
        for _ in range(12 if lead == "auto" else 4):
            got, gutil, grun = c.dispatch(tk)
            st = c.stats()
            seen.append((st["zone_rows"], st["rounds"]))
            assert np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil), seen
        if lead == "off":
            assert all(z == 0 for z, _ in seen), seen
        elif lead == "auto":
            assert all(x == (0, 4) for x in seen[:4]), seen           # the chain, seen three times behind a cold batch
            assert all(z >= 2 for z, _ in seen[4:7]), seen            # ... the walk, tried three times
            assert all(z >= 2 and r <= 2 for z, r in seen[7:]), seen  # ... and kept: it costs less
        else:
            assert all(z >= 2 for z, _ in seen), seen   # the stretch spans several chunks
            assert seen[-1][1] <= 2, seen                # ... which come out consistent in their first replay
            if lead:
                assert seen[0][1] > 2, seen              # (not with a walk that starts inside the transient)
    finally:
        c.close()


def test_walk_in_pipelined_committing_batches(monkeypatch):
    """The walk inside ydc_dispatch_device_async batches that COMMIT: the dedicated tier runs out
    in the second of three batches of 300k requests (cfg3's registry), every batch is enqueued
    while the one before it is still running. Three batches == the oracle on their concatenation."""
    monkeypatch.setenv("YDC_ZONE_GUESS", "2")
    DA = binding.DeviceArray
    sv, tk = synth.make_config("cfg3")
    per, nb = 300_000, 3
    tk = {k: v[:per * nb] for k, v in tk.items()}
    want, _, wrun = O.dispatch(sv, tk, "sorted")
    S = len(sv["version"])
    c = binding.Context(device=0)
    try:
        c.upload_servants(pack.to_abi_columns(sv))
        cols = [[DA.from_numpy(tk[k][b * per:(b + 1) * per]) for k in ("env_id", "min_version", "requestor_ip")]
                for b in range(nb)]
        outs = [DA.from_numpy(np.full(per, 0xDEADBEEF, np.uint32)) for _ in range(nb)]
        runs = [DA(S, np.uint32) for _ in range(nb)]
        served = []
        c.dispatch_device_async(*cols[0], outs[0], None, runs[0], commit=True)
        for b in range(1, nb):
            c.dispatch_device_async(*cols[b], outs[b], None, runs[b], commit=True)
            c.dispatch_wait()
            served.append(c.stats()["zone_rows"])
        c.dispatch_wait()
        served.append(c.stats()["zone_rows"])
        got = np.concatenate([o.numpy() for o in outs])
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]], served)
        assert np.array_equal(runs[-1].numpy(), wrun) and np.array_equal(c.get_running(), wrun)
        assert served[1] >= 2 and served[0] == 0 and served[2] == 0, served  # (the tier ends in the second batch)
    finally:
        c.close()


def test_walk_on_random_pools(monkeypatch):
    """The walk of the tier's end (zone_guess.h) forced on (YDC_ZONE_GUESS=2, radix pipeline) over
    random pools of 4 digests: batches that end before the dedicated tier does, reach just past it,
    or start with it exhausted; hosts with several servants, oversubscription, initial load,
    committed halves. Every placement, utilisation and running_tasks column equal to the oracle's."""
    monkeypatch.setenv("YDC_ZONE_GUESS", "2")
    monkeypatch.setenv("YDC_BINSORT", "0")
    walked = 0
    for seed in range(300, 324):
        rng = np.random.default_rng(seed)
        kw = dict(seed=seed, n_tasks=int(rng.choice([40_000, 70_000, 130_000, 220_000])),
                  n_servants=int(rng.choice([600, 1500, 4000])), n_envs=4,
                  self_frac=float(rng.choice([0.0, 0.1, 0.4])), unknown_env_frac=float(rng.choice([0.0, 0.01])),
                  min_version_20_frac=float(rng.choice([0.0, 0.5])))
        if rng.random() < 0.3:
            kw["oversubscribed"] = True
        if rng.random() < 0.3:
            kw["shared_ip_frac"] = 0.2
        if rng.random() < 0.4:
            kw["initial_running"] = True
        monkeypatch.setenv("YDC_CHUNK_SIZE", str(int(rng.choice([64, 128, 512]))))
        sv, tk = cases.random_case(**kw)
        n = len(tk["env_id"])
        want, wutil, wrun = O.dispatch(sv, tk, "sorted")
        c = binding.Context(device=0)
        try:
            c.upload_servants(pack.to_abi_columns(sv))
            if rng.random() < 0.4:
                cut = int(rng.integers(1, n))
                a, ua, _ = c.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
                walked += c.stats()["zone_rows"] >= 2
                b, ub, grun = c.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
                got, gutil = np.concatenate([a, b]), np.concatenate([ua, ub])
            else:
                got, gutil, grun = c.dispatch(tk)
            st = c.stats()
            walked += st["zone_rows"] >= 2
        finally:
            c.close()
        assert np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil), (kw, st)
    assert walked >= 4, walked  # (the shapes are drawn so that the tier ends inside a good part of the batches)


@pytest.mark.parametrize("counted", ["1", "0"])
def test_long_release_lists(counted, monkeypatch):
    """FreeTask for a whole batch's grants at once (ydc_release_slots / _device): long lists are
    counted per servant in LDS first and go out as one atomic per servant (release_counted=0: an
    atomic per slot). COMMIT + release restores the registry; a second batch places like the first."""
    monkeypatch.setenv("YDC_RELEASE_COUNTED", counted)
    DA = binding.DeviceArray
    for cfg, n_srv in (("cfg2", None), ("cfg3", 14_000)):
        sv, tk = synth.make_config(cfg, n_tasks=150_000, n_servants=n_srv)
        c = binding.Context(device=0)
        try:
            c.upload_servants(pack.to_abi_columns(sv))
            before = c.get_running().copy()
            d = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
            out = DA(len(tk["env_id"]), np.uint32)
            c.dispatch_device(d[0], d[1], d[2], out, commit=True)
            first = out.numpy().copy()
            granted = first[first < binding.IDX_ENV_NOT_FOUND]
            assert np.array_equal(c.get_running() - before, np.bincount(granted, minlength=len(before)))
            c.release_slots_device(out)      # (entries that are no servant index are skipped)
            assert np.array_equal(c.get_running(), before)
            c.dispatch_device(d[0], d[1], d[2], out, commit=True)
            assert np.array_equal(out.numpy(), first)
            c.release_slots(granted)         # the same list from the host
            assert np.array_equal(c.get_running(), before)
        finally:
            c.close()
