"""Pins the oracle: the reference's own golden vectors
(yadcc/scheduler/task_dispatcher_test.cc:29-144,146-186,216-298,
running_task_bookkeeper_test.cc:24-42) replayed (a) through the reference's own
translation units compiled verbatim (oracle/_ref) with a manual clock instead of
the tests' real sleeps, and (b) through the plain-C restatement; then the two
against each other on random pools."""
import numpy as np
import pytest

from oracle import oraclebind as O
from oracle import refbind as R
from tests import cases

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
G50 = 50 << 30


@needs_ref
def test_ref_golden_all():
    """task_dispatcher_test.cc:29-144 (`All`), sleeps replaced by clock + timer."""
    d = R.RefDispatcher()
    d.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, memory_available=G50)
    st, _, _ = d.wait_for_starting_new_task("127.0.0.1", "not found", timeout_in_ms=1000)
    assert st == R.ENV_NOT_FOUND
    tasks = []
    for _ in range(10):
        st, tid, loc = d.wait_for_starting_new_task("127.0.0.1", "digest", expires_in_ms=5000,
                                                    timeout_in_ms=1000)
        assert st == R.OK and loc == "127.0.0.1:1234"
        tasks.append(tid)
    st, _, _ = d.wait_for_starting_new_task("127.0.0.1", "digest", timeout_in_ms=1000)
    assert st == R.TIMEOUT
    assert not d.keep_task_alive(12345678, 1000)
    for t in tasks:
        assert d.keep_task_alive(t, 1000)
    assert d.notify_servant_running_tasks("127.0.0.1:1234", [tasks[0], 1000002, 1000003]) == [
        1000002, 1000003]
    R.clock_advance_ms(2000)
    R.fire_timers()  # leases (1 s) expired -> zombies
    for t in tasks:
        assert not d.keep_task_alive(t, 1000)
    assert d.notify_servant_running_tasks("127.0.0.1:1234", tasks) == tasks
    d.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, memory_available=G50,
                         expires_in_ms=1000)
    R.clock_advance_ms(2000)
    R.fire_timers()  # servant expired
    st, _, _ = d.wait_for_starting_new_task("127.0.0.1", "digest", timeout_in_ms=1000)
    assert st != R.OK
    d.close()


@needs_ref
def test_ref_golden_prefer_dedicated():
    """task_dispatcher_test.cc:146-186."""
    d = R.RefDispatcher()
    d.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, priority=2,
                         memory_available=G50)
    st, tid, loc = d.wait_for_starting_new_task("127.0.0.1", "digest")
    assert (st, loc) == (R.OK, "127.0.0.1:1234")  # self allowed when alone
    d.free_task(tid)
    d.keep_servant_alive("192.168.0.1:1234", ["digest"], 10, 10, 2, priority=1,
                         memory_available=G50)
    st, tid, loc = d.wait_for_starting_new_task("127.0.0.1", "digest")
    assert (st, loc) == (R.OK, "192.168.0.1:1234")
    d.close()


LB = [("192.168.0.0:0000", 7, 16, 16), ("192.168.0.1:1111", 7, 16, 1),
      ("192.168.0.2:2222", 8, 16, 5), ("192.168.0.3:3333", 6, 16, 12)]
LB_EXPECT = [1, 2, 3, 2, 1, 2, 3]  # task_dispatcher_test.cc:237-297


@needs_ref
def test_ref_golden_load_balance():
    """task_dispatcher_test.cc:216-298."""
    d = R.RefDispatcher()
    loads = {}
    loc0, mt, npz, ld = LB[0]
    d.keep_servant_alive(loc0, ["Load Balance"], mt, npz, ld, memory_available=G50)
    st, _, _ = d.wait_for_starting_new_task("127.0.0.3", "Load Balance", timeout_in_ms=1000)
    assert st == R.TIMEOUT  # overloaded servant: Timeout, not EnvironmentNotFound (:217-228)
    for loc, mt, npz, ld in LB[1:]:
        d.keep_servant_alive(loc, ["Load Balance"], mt, npz, ld, memory_available=G50)
        loads[loc] = [mt, npz, ld]
    got = []
    for _ in LB_EXPECT:
        st, _, loc = d.wait_for_starting_new_task("127.0.0.3", "Load Balance")
        assert st == R.OK
        got.append([x[0] for x in LB].index(loc))
        loads[loc][2] += 1  # the test re-heartbeats the chosen servant with load + 1
        d.keep_servant_alive(loc, ["Load Balance"], *loads[loc], memory_available=G50)
    assert got == LB_EXPECT
    d.close()


def _lb_columns(loads, running):
    n = len(LB)
    return {
        "version": np.full(n, 8, np.uint32),
        "num_processors": np.array([x[2] for x in LB], np.uint32),
        "current_load": np.array(loads, np.uint32),
        "max_tasks": np.array([x[1] for x in LB], np.uint32),
        "running_tasks": np.array(running, np.uint32),
        "priority": np.full(n, 2, np.uint32),
        "total_memory": np.zeros(n, np.uint64),  # the reference test never sets it
        "memory_available": np.full(n, G50, np.uint64),
        "env_mask": np.ones(n, np.uint64),
        "ip": np.array([0xC0A80000 + i for i in range(n)], np.uint32),
        "port": np.array([0, 1111, 2222, 3333], np.uint32),
    }


@pytest.mark.parametrize("method", ["scan", "sorted"])
def test_restatement_golden_load_balance(method):
    """Same vector through the plain-C restatement, one request per batch, re-heartbeating
    load + 1 in between exactly like the reference test does."""
    loads = [x[3] for x in LB]
    running = [0] * len(LB)
    one = {"env_id": np.zeros(1, np.uint32), "min_version": np.full(1, 8, np.uint32),
           "requestor_ip": np.array([0x7F000003], np.uint32)}
    # overloaded servant alone: Timeout
    sv0 = {k: v[:1] for k, v in _lb_columns(loads, running).items()}
    idx, _, _ = O.dispatch(sv0, one, method)
    assert idx[0] == O.IDX_TIMEOUT
    got = []
    for _ in LB_EXPECT:
        idx, util, run = O.dispatch(_lb_columns(loads, running), one, method)
        s = int(idx[0])
        got.append(s)
        running = run.tolist()
        loads[s] += 1
    assert got == LB_EXPECT


@pytest.mark.parametrize("method", ["scan", "sorted"])
def test_restatement_golden_all_and_dedicated(method):
    # `All` (:29-88): unknown digest -> EnvNotFound; 10 grants on the requestor's own
    # (only) servant; 11th -> Timeout.
    sv = {"version": [8], "num_processors": [10], "current_load": [0], "max_tasks": [10],
          "running_tasks": [0], "priority": [2], "total_memory": [0],
          "memory_available": [G50], "env_mask": [1], "ip": [0x7F000001], "port": [1234]}
    tk = {"env_id": [9999] + [0] * 11, "min_version": [8] * 12, "requestor_ip": [0x7F000001] * 12}
    idx, _, run = O.dispatch(sv, tk, method)
    assert idx.tolist() == [O.IDX_ENV_NOT_FOUND] + [0] * 10 + [O.IDX_TIMEOUT]
    assert run.tolist() == [10]
    # PreferDedicated (:146-186)
    sv = {"version": [8, 8], "num_processors": [10, 10], "current_load": [0, 2],
          "max_tasks": [10, 10], "running_tasks": [0, 0], "priority": [2, 1],
          "total_memory": [0, 0], "memory_available": [G50, G50], "env_mask": [1, 1],
          "ip": [0x7F000001, 0xC0A80001], "port": [1234, 1234]}
    tk = {"env_id": [0], "min_version": [8], "requestor_ip": [0x7F000001]}
    idx, _, _ = O.dispatch(sv, tk, method)
    assert idx.tolist() == [1]


@needs_ref
def test_ref_running_task_bookkeeper():
    """running_task_bookkeeper_test.cc:24-42 through TaskDispatcher's forwarding methods
    (NotifyServantRunningTasks -> SetServantRunningTasks, task_dispatcher.cc:274-275)."""
    d = R.RefDispatcher()
    d.keep_servant_alive("10.0.0.1:1", ["x"], 10, 10, 0, memory_available=G50)
    grants = []
    for _ in range(3):
        st, tid, _ = d.wait_for_starting_new_task("1.1.1.1", "x", expires_in_ms=5000)
        grants.append(tid)
    assert d.notify_servant_running_tasks("10.0.0.1:1", grants, [100, 101, 102]) == []
    assert d.get_running_tasks() == list(zip([100, 101, 102], grants))
    d.keep_servant_alive("10.0.0.1:1", ["x"], 10, 10, 0, memory_available=G50, expires_in_ms=1)
    R.clock_advance_ms(1000)
    R.fire_timers()  # servant expires -> DropServant (task_dispatcher.cc:509-510)
    assert d.get_running_tasks() == []
    d.close()


def test_parse_size():
    """yadcc/common/parse_size_test.cc values + the flag default."""
    assert O.try_parse_size("10G") == 10 << 30
    assert O.try_parse_size("1") == 1 and O.try_parse_size("1B") == 1
    assert O.try_parse_size("1K") == 1024 and O.try_parse_size("2M") == 2 << 20
    assert O.try_parse_size("1.5G") is None and O.try_parse_size("G") is None
    from yadcc_amd import pack
    for s in ("10G", "1", "1B", "1K", "2M", "1.5G", "G", "12X"):
        assert pack.parse_size(s) == O.try_parse_size(s)


@needs_ref
@pytest.mark.parametrize("name,kw", cases.SMALL_CASES, ids=[c[0] for c in cases.SMALL_CASES])
def test_restatement_equals_verbatim_reference(name, kw):
    sv, tk = cases.random_case(**kw)
    d = R.RefDispatcher()
    d.load_servants(sv)
    ref_idx, _, _, _ = d.dispatch_batch(tk)
    d.close()
    for method in ("scan", "sorted"):
        idx, util, run = O.dispatch(sv, tk, method)
        assert np.array_equal(idx, ref_idx), method
    a = O.dispatch(sv, tk, "scan")
    b = O.dispatch(sv, tk, "sorted")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@needs_ref
@pytest.mark.parametrize("name,sv,tk", cases.handmade_cases(),
                         ids=[c[0] for c in cases.handmade_cases()])
def test_handmade_equals_verbatim_reference(name, sv, tk):
    # (huge_capacity — capacities >= 2^21, the fp64-key path — presets its millions of running
    # tasks through the FRIEND_TEST door of oracle/ref_driver.cc instead of priming them)
    d = R.RefDispatcher()
    d.load_servants(sv)
    ref_idx, _, _, _ = d.dispatch_batch(tk)
    d.close()
    for method in ("scan", "sorted"):
        idx, _, _ = O.dispatch(sv, tk, method)
        assert np.array_equal(idx, ref_idx), method


@needs_ref
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_fp64_key_pools_equal_verbatim_reference(seed):
    """Capacities >= 2^21 (the fp64-key path): both restatements against the reference itself,
    whose running_tasks are preset through the FRIEND_TEST door of oracle/ref_driver.cc."""
    sv, tk = cases.huge_capacity_pool(seed=seed)
    d = R.RefDispatcher()
    d.load_servants(sv)
    ref_idx, _, _, _ = d.dispatch_batch(tk)
    d.close()
    assert (ref_idx < R.IDX_ENV_NOT_FOUND).sum() > 500 and (ref_idx == R.IDX_TIMEOUT).sum() > 0
    for method in ("scan", "sorted"):
        idx, _, _ = O.dispatch(sv, tk, method)
        assert np.array_equal(idx, ref_idx), method


def test_capacity_formula_corners():
    """GetCapacityAvailable, task_dispatcher.cc:283-313."""
    cap = O.capacity_available
    assert cap(16, 16, 7, 0, G50, 0) == 0          # fully loaded by others
    assert cap(16, 12, 6, 0, G50, 0) == 4          # LoadBalanceCase servant3: 0 / 4
    assert cap(16, 12, 6, 0, G50, 2) == 6          # grows with running while running <= load
    assert cap(16, 3, 8, 0, G50, 5) == 8           # load < running: no foreign load
    assert cap(16, 40, 8, 0, G50, 0) == 0          # load >= nproc clamps at 0
    assert cap(16, 0, 8, 64 << 30, 1 << 30, 3) == 3  # low memory: capacity == running
    assert cap(16, 0, 8, 0, 1 << 30, 3) == 8       # memory not reported: ignored
    assert cap(16, 0, 0, 0, G50, 0) == 0
