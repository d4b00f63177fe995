"""The small-batch path (yadcc_amd/csrc/tick_kernel.h, ydc_dispatch_tick): one launch places a
handful of requests with the reference's own arg-min (task_dispatcher.cc:362-451), heartbeat rows
and released grants riding in the same launch. Checked against the literal restatement of the
reference (oracle "scan") — bit-exact placement, running_tasks and utilisation — on the shared
case families forced through it (YDC_SMALL_BATCH), on randomised small shapes, on registries at
every servants-per-thread width of the kernel, and as sequences of ticks whose deltas are replayed
on the oracle's snapshot."""
import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from yadcc_amd import binding, pack, synth

pytestmark = pytest.mark.gpu


def check(ctx, sv, tk, method="scan", **kw):
    want, wutil, wrun = O.dispatch(sv, tk, method)
    ctx.upload_servants(pack.to_abi_columns(sv))
    got, gutil, grun = ctx.dispatch(tk, **kw)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at task %d: gpu %d oracle %d (%d total) stats=%s" % (
        bad[0], got[bad[0]], want[bad[0]], bad.size, ctx.stats())
    assert np.array_equal(grun, wrun)
    assert np.array_equal(gutil, wutil)
    st = ctx.stats()
    assert st["granted"] == int((want < O.IDX_ENV_NOT_FOUND).sum())
    assert st["timeouts"] == int((want == O.IDX_TIMEOUT).sum())
    assert st["env_not_found"] == int((want == O.IDX_ENV_NOT_FOUND).sum())
    return st


@pytest.fixture(params=["packed", "fp64"])
def forced(monkeypatch, request):
    """A context whose every batch takes the one-workgroup kernel, whatever its size — with the
    one-word candidate (integer key above the registry index, where capacities are below 2^10 and
    both fit 32 bits) and with the reference's double as the key."""
    monkeypatch.setenv("YDC_SMALL_BATCH", "100000000")
    monkeypatch.setenv("YDC_PACKED_TICK", "1" if request.param == "packed" else "0")
    c = binding.Context(device=0)
    yield c
    c.close()


@pytest.fixture()
def never(monkeypatch):
    monkeypatch.setenv("YDC_SMALL_BATCH", "0")
    c = binding.Context(device=0)
    yield c
    c.close()


def took_the_tick_kernel(st):
    return st["small_batch"] == 1


@pytest.mark.parametrize("name,kw", cases.SMALL_CASES, ids=[c[0] for c in cases.SMALL_CASES])
def test_small_cases_forced_through_the_tick_kernel(forced, name, kw):
    sv, tk = cases.random_case(**kw)
    st = check(forced, sv, tk)
    assert took_the_tick_kernel(st), st


@pytest.mark.parametrize("name,sv,tk", cases.handmade_cases(), ids=[c[0] for c in cases.handmade_cases()])
def test_handmade_cases_both_ways(forced, never, name, sv, tk):
    st = check(forced, sv, tk)
    assert len(tk["env_id"]) == 0 or took_the_tick_kernel(st), st
    st = check(never, sv, tk)
    assert len(tk["env_id"]) == 0 or not took_the_tick_kernel(st), st


def test_default_threshold_routes_by_batch_size():
    c = binding.Context(device=0)
    sv, tk = cases.random_case(seed=21, n_tasks=64, n_servants=300, n_envs=3, self_frac=0.3)
    assert took_the_tick_kernel(check(c, sv, tk))
    sv, tk = cases.random_case(seed=22, n_tasks=65, n_servants=300, n_envs=3, self_frac=0.3)
    assert not took_the_tick_kernel(check(c, sv, tk))
    c.close()


@pytest.mark.parametrize("n_servants,n_tasks", [(1023, 64), (1025, 40), (2000, 1), (2000, 16), (3000, 64),
                                                (4097, 64), (8000, 33), (8193, 64), (16000, 64), (16384, 200)])
def test_every_register_width(forced, n_servants, n_tasks):
    """1, 2, 4 (columns in registers), 8, 16 (re-read by the winner) servants per thread; more than
    64 requests = the pointer path (columns staged in page-locked memory)."""
    sv, tk = cases.random_case(seed=300 + n_servants % 97, n_tasks=n_tasks, n_servants=n_servants, n_envs=4,
                               self_frac=0.3, shared_ip_frac=0.05, unknown_env_frac=0.02)
    # (a pool sized for the batch would leave most servants idle: start it half full)
    rng = np.random.default_rng(n_servants)
    top = np.minimum(sv["max_tasks"], sv["num_processors"]).astype(np.int64)
    sv["running_tasks"] = (rng.random(n_servants) * (top + 2)).astype(np.uint32)
    st = check(forced, sv, tk)
    assert took_the_tick_kernel(st), st


def test_beyond_the_kernels_registry_limits_falls_back(forced):
    sv, tk = cases.random_case(seed=41, n_tasks=50, n_servants=16385, n_envs=2)
    st = check(forced, sv, tk)
    assert not took_the_tick_kernel(st), st


def test_many_classes_and_wide_masks(forced):
    """150 digests (3 mask words per servant), about one class per servant: an eligible-class mask
    of many words in LDS; and past 4096 classes the batch pipeline takes over."""
    sv, tk = cases.random_case(seed=12, n_tasks=300, n_servants=900, n_envs=150, unknown_env_frac=0.02, self_frac=0.2)
    st = check(forced, sv, tk)
    assert st["n_classes"] > 256 and took_the_tick_kernel(st), st
    sv, tk = cases.random_case(seed=13, n_tasks=60, n_servants=3000, n_envs=150, self_frac=0.2)
    st = check(forced, sv, tk)
    assert st["n_classes"] > 700 and took_the_tick_kernel(st), st
    # every servant its own version: one class per servant
    sv, tk = cases.random_case(seed=14, n_tasks=60, n_servants=4500, n_envs=3, self_frac=0.2)
    sv["version"] = (20 + np.arange(4500)).astype(np.uint32)
    sv["num_processors"][:] = 16
    sv["max_tasks"][:] = 6
    st = check(forced, sv, tk)
    assert st["n_classes"] > 4096 and not took_the_tick_kernel(st), st


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_small_shapes(forced, seed):
    """Random tiny shapes — where every pick moves the minimum, own-host requests, shared hosts,
    initial load: 350 cases per seed against the literal restatement."""
    rng = np.random.default_rng(1000 + seed)
    for it in range(350):
        n_servants = int(rng.integers(1, 40)) if it % 3 else int(rng.integers(40, 1500))
        n_tasks = int(rng.integers(1, 130))
        kw = dict(seed=int(rng.integers(1 << 30)), n_tasks=n_tasks, n_servants=n_servants,
                  n_envs=int(rng.integers(1, 6)), self_frac=float(rng.choice([0.0, 0.3, 1.0])),
                  shared_ip_frac=float(rng.choice([0.0, 0.3, 0.8])),
                  unknown_env_frac=float(rng.choice([0.0, 0.05])),
                  oversubscribed=bool(rng.integers(2)), initial_running=bool(rng.integers(2)))
        sv, tk = cases.random_case(**kw)
        try:
            check(forced, sv, tk)
        except AssertionError as e:
            raise AssertionError("case %r: %s" % (kw, e))


def _rows_of(sv, idx, flags):
    rows = np.zeros(len(idx), binding.ROW_DTYPE)
    for k, col in (("version", "version"), ("num_processors", "num_processors"), ("current_load", "current_load"),
                   ("max_tasks", "max_tasks"), ("ip_id", "ip")):
        rows[k] = np.asarray(sv[col])[idx]
    rows["flags"] = flags[idx]
    em = np.asarray(sv["env_mask"])
    rows["env_mask"] = em[idx] if em.ndim == 1 else em[idx, 0]
    return rows


@pytest.mark.parametrize("packed", ["1", "0"])
@pytest.mark.parametrize("resident", ["1", "0"])
@pytest.mark.parametrize("n_servants,n_envs,big", [(60, 1, False), (700, 3, False), (2500, 4, True), (9000, 2, True),
                                                   (400, 150, False)])
def test_ticks_with_heartbeats_and_releases(n_servants, n_envs, big, resident, packed, monkeypatch):
    """Sequences of scheduler turns: heartbeats (new load / memory / capacity figures — and now and
    then another version or environment set, which changes structure and takes the general path),
    released grants, a handful of requests — COMMITted; the oracle replays every turn on its own
    snapshot. Delta lists beyond what travels as kernel arguments (16 rows, 64 releases) included.
    resident = 1: the kernel of a turn stays on its CU and takes the following turns from its
    mailbox (until a turn it does not take — a structural heartbeat, a long list — ends it);
    resident = 0: every turn is a launch."""
    monkeypatch.setenv("YDC_RESIDENT", resident)
    monkeypatch.setenv("YDC_PACKED_TICK", packed)
    rng = np.random.default_rng(77 + n_servants)
    sv, _ = cases.random_case(seed=5 + n_servants, n_tasks=3 * n_servants, n_servants=n_servants, n_envs=n_envs,
                              shared_ip_frac=0.1)
    sv = {k: np.array(v, copy=True) for k, v in sv.items()}
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    held = []  # servant index of every live grant
    fast = slow = 0
    for turn in range(120):
        # heartbeats
        n_upd = int(rng.choice([0, 1, 3, 16, 40])) if big else int(rng.choice([0, 1, 2, 5]))
        idx = np.sort(rng.choice(n_servants, size=min(n_upd, n_servants), replace=False)).astype(np.uint32)
        for s in idx:
            sv["current_load"][s] = rng.integers(0, int(sv["num_processors"][s]) + 3)
            if rng.random() < 0.2:
                sv["memory_available"][s] = rng.integers(1 << 30, 40 << 30)
            if rng.random() < 0.1:  # capacity figures that keep min(max_tasks, nproc): no structure changes
                keep = min(int(sv["max_tasks"][s]), int(sv["num_processors"][s]))
                if int(sv["max_tasks"][s]) == keep and keep:
                    sv["num_processors"][s] = keep + int(rng.integers(0, 9))
            if rng.random() < 0.03:
                sv["version"][s] = 19 + int(rng.integers(0, 3))  # structural
            if rng.random() < 0.02 and n_envs > 1 and np.asarray(sv["env_mask"]).ndim == 1:
                sv["env_mask"][s] = np.uint64(rng.integers(1, 1 << n_envs))  # structural
        flags = pack.servant_flags(sv)
        rows = _rows_of(sv, idx, flags)
        em = np.asarray(sv["env_mask"])
        env_masks = em[idx] if em.ndim == 2 else None
        # releases
        n_rel = min(len(held), int(rng.choice([0, 1, 4, 64, 130])) if big else int(rng.choice([0, 1, 3, 9])))
        rel = []
        for _ in range(n_rel):
            rel.append(held.pop(int(rng.integers(len(held)))))
        for s in rel:
            sv["running_tasks"][s] -= 1
        # requests
        n = int(rng.choice([0, 1, 2, 7, 16, 64]))
        tk = synth.make_tasks(n, sv, n_envs=n_envs, seed=int(rng.integers(1 << 30)), self_frac=0.3,
                              unknown_env_frac=0.02)
        want, wutil, wrun = O.dispatch(sv, tk, "scan")
        got, gutil = c.dispatch_tick(tk, idx, rows, rel, env_masks=env_masks, want_util=True)
        assert np.array_equal(got, want), (turn, got, want, c.stats())
        assert np.array_equal(gutil, wutil), turn
        st = c.stats()
        if n:
            fast += took_the_tick_kernel(st)
            slow += not took_the_tick_kernel(st)
        sv["running_tasks"] = wrun
        assert np.array_equal(c.get_running(), wrun), turn
        held.extend(int(s) for s in got if s < O.IDX_ENV_NOT_FOUND)
    c.close()
    assert fast > 40, (fast, slow)


def test_dispatch_device_small_batch_caller_owned_outputs():
    """ydc_dispatch_device with a handful of requests: columns and outputs in caller-owned device
    buffers, no COMMIT (running_tasks after the batch in the caller's buffer, the resident column
    untouched), then COMMIT."""
    DA = binding.DeviceArray
    c = binding.Context(device=0)
    sv, tk = cases.random_case(seed=55, n_tasks=48, n_servants=2000, n_envs=3, self_frac=0.3, initial_running=True)
    n, S = 48, 2000
    c.upload_servants(pack.to_abi_columns(sv))
    cols = [DA.from_numpy(tk[k]) for k in ("env_id", "min_version", "requestor_ip")]
    d_idx = DA.from_numpy(np.full(n, 0xDEADBEEF, np.uint32))
    d_util = DA.from_numpy(np.full(n, -7.0, np.float64))
    d_run = DA.from_numpy(np.full(S, 0xDEADBEEF, np.uint32))
    want, wutil, wrun = O.dispatch(sv, tk, "scan")
    for rep in range(2):
        c.dispatch_device(cols[0], cols[1], cols[2], d_idx, d_util, d_run)
        assert took_the_tick_kernel(c.stats())
        assert np.array_equal(d_idx.numpy(), want) and np.array_equal(d_util.numpy(), wutil)
        assert np.array_equal(d_run.numpy(), wrun)
        assert np.array_equal(c.get_running(), np.asarray(sv["running_tasks"], np.uint32))
    c.dispatch_device(cols[0], cols[1], cols[2], d_idx, None, None, commit=True)
    assert np.array_equal(d_idx.numpy(), want) and np.array_equal(c.get_running(), wrun)
    c.close()


def test_tick_then_batch_then_tick_share_the_registry():
    """The one-launch path and the batch pipeline work on the same resident columns: a tick's
    COMMIT is what the next large batch sees, and the other way round."""
    c = binding.Context(device=0)
    sv, _ = cases.random_case(seed=66, n_tasks=9000, n_servants=500, n_envs=2)
    sv = {k: np.array(v, copy=True) for k, v in sv.items()}
    c.upload_servants(pack.to_abi_columns(sv))
    rng = np.random.default_rng(9)
    for turn in range(12):
        n = 20 if turn % 2 == 0 else 3000
        tk = synth.make_tasks(n, sv, n_envs=2, seed=int(rng.integers(1 << 30)), self_frac=0.2)
        want, _, wrun = O.dispatch(sv, tk, "sorted")
        got, _, _ = c.dispatch(tk, commit=True, want_util=False, want_running=False)
        assert np.array_equal(got, want), turn
        assert took_the_tick_kernel(c.stats()) == (n == 20)
        sv["running_tasks"] = wrun
        assert np.array_equal(c.get_running(), wrun)
    c.close()


def test_resident_kernel_idle_exit_and_relaunch(monkeypatch):
    """The resident kernel leaves by itself when nobody has asked for a while; the next turn launches
    a new one — also when the command and the exit cross (turns spaced around the idle time)."""
    import time
    monkeypatch.setenv("YDC_RESIDENT_IDLE_MS", "2")
    rng = np.random.default_rng(5)
    sv, _ = cases.random_case(seed=77, n_tasks=20000, n_servants=1500, n_envs=3)
    sv = {k: np.array(v, copy=True) for k, v in sv.items()}
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    held = []
    for turn in range(300):
        rel = [held.pop(int(rng.integers(len(held)))) for _ in range(min(len(held), int(rng.integers(0, 4))))]
        for s in rel:
            sv["running_tasks"][s] -= 1
        tk = synth.make_tasks(int(rng.integers(1, 9)), sv, n_envs=3, seed=int(rng.integers(1 << 30)), self_frac=0.2)
        want, _, wrun = O.dispatch(sv, tk, "scan")
        got, _ = c.dispatch_tick(tk, release_idx=rel)
        assert np.array_equal(got, want), turn
        sv["running_tasks"] = wrun
        held.extend(int(s) for s in got if s < O.IDX_ENV_NOT_FOUND)
        time.sleep(float(rng.choice([0.0, 0.0015, 0.002, 0.0025, 0.01])))
    assert np.array_equal(c.get_running(), sv["running_tasks"])
    c.close()


def test_resident_kernel_and_everything_else_interleaved():
    """Every other entry point ends the resident kernel first and sees what it left in HBM: large
    batches, device-pointer batches, row updates, releases, get / set of running_tasks, a removed
    servant — interleaved with resident turns, each step against the oracle's snapshot."""
    rng = np.random.default_rng(8)
    sv, _ = cases.random_case(seed=78, n_tasks=30000, n_servants=800, n_envs=3, shared_ip_frac=0.05)
    sv = {k: np.array(v, copy=True) for k, v in sv.items()}
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    for turn in range(60):
        tk = synth.make_tasks(int(rng.integers(1, 30)), sv, n_envs=3, seed=int(rng.integers(1 << 30)), self_frac=0.2)
        want, _, wrun = O.dispatch(sv, tk, "scan")
        got, _ = c.dispatch_tick(tk)
        assert np.array_equal(got, want), turn
        sv["running_tasks"] = wrun
        what = turn % 6
        if what == 0:  # a large committed batch through the pipeline
            tk = synth.make_tasks(2000, sv, n_envs=3, seed=int(rng.integers(1 << 30)), self_frac=0.2)
            want, _, wrun = O.dispatch(sv, tk, "sorted")
            got, _, _ = c.dispatch(tk, commit=True, want_util=False, want_running=False)
            assert np.array_equal(got, want), turn
            sv["running_tasks"] = wrun
        elif what == 1:
            assert np.array_equal(c.get_running(), sv["running_tasks"])
        elif what == 2:  # released grants through the plain entry point
            busy = np.nonzero(sv["running_tasks"] > 0)[0]
            rel = rng.choice(busy, size=min(5, len(busy)), replace=False).astype(np.uint32)
            c.release_slots(rel)
            sv["running_tasks"][rel] -= 1
        elif what == 3:  # a structural heartbeat through the plain entry point
            s = int(rng.integers(len(sv["version"])))
            sv["version"][s] = 19 + int(rng.integers(0, 3))
            c.update_servants([s], _rows_of(sv, np.array([s]), pack.servant_flags(sv)))
        elif what == 4:  # a servant expires
            s = int(rng.integers(len(sv["version"])))
            c.remove_servants([s])
            sv = {k: np.delete(v, s, axis=0) for k, v in sv.items()}
        else:
            c.set_running(sv["running_tasks"])
    assert np.array_equal(c.get_running(), sv["running_tasks"])
    c.close()


def test_resident_kernel_alternating_utilisation():
    """One context, COMMITting ticks that alternately want and do not want the chosen servants'
    utilisation (round-5 advisor finding: a resident kernel fixes its utilisation output at launch;
    a call of the other kind must not be answered from a kernel that never stores it). Small and
    7+ request batches, every answer and every utilisation against the oracle."""
    c = binding.Context(device=0)
    sv, _ = cases.random_case(seed=77, n_tasks=10, n_servants=600, n_envs=2)
    sv = {k: np.array(v, copy=True) for k, v in sv.items()}
    c.upload_servants(pack.to_abi_columns(sv))
    rng = np.random.default_rng(17)
    resident_before = c.stats()["tick_resident_calls"]
    for turn in range(60):
        n = int(rng.choice([1, 3, 7, 8, 20]))
        tk = synth.make_tasks(n, sv, n_envs=2, seed=int(rng.integers(1 << 30)), self_frac=0.2)
        want, wutil, wrun = O.dispatch(sv, tk, "scan")
        want_util = bool((turn // 2) % 2) if turn < 40 else bool(turn % 2)
        got, gutil = c.dispatch_tick(tk, commit=True, want_util=want_util)
        assert np.array_equal(got, want), (turn, n, want_util)
        if want_util:
            assert np.array_equal(gutil, wutil), (turn, n)
        assert took_the_tick_kernel(c.stats())
        sv["running_tasks"] = wrun
    assert np.array_equal(c.get_running(), sv["running_tasks"])
    # (pairs of equal calls in the first 40 turns: the second of a pair is answered by the kernel the
    # first one left resident)
    assert c.stats()["tick_resident_calls"] - resident_before >= 15
    c.close()
