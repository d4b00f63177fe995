"""Host logic of GpuTaskDispatcher that needs no placement (runs without a GPU): servant
registry, expiry, lease / unknown-id answers, running-task bookkeeping, DumpInternals keys —
against the reference class itself (oracle/_ref) where it is built — and the rule that a
dispatcher without a device fails every wait loudly instead of placing on the CPU."""
import pytest

from oracle import refbind as R
from yadcc_amd import binding, dispatcher as D

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
G50 = 50 << 30


@pytest.fixture
def td():
    d = D.GpuTaskDispatcher(device=-1)
    yield d
    d.close()


def test_td_symbols_exported():
    L = binding.lib()
    for s in D.TD_SYMBOLS:
        assert hasattr(L, s), s


def test_wait_without_device_fails_loudly(td):
    assert td.device_status == -3  # YDC_ERR_NO_DEVICE
    td.keep_servant_alive("10.0.0.1:8335", ["d"], 8, 16, 0)
    with pytest.raises(binding.YdcError, match="no usable gfx950 device"):
        td.wait_for_starting_new_task("10.0.0.9", "d")
    with pytest.raises(binding.YdcError, match="no usable gfx950 device"):
        td.wait_for_starting_new_tasks(["10.0.0.9"], ["d"], [0])


def test_registry_upsert_expiry_and_dump(td):
    """KeepServantAlive appends in registration order, a renewal replaces the personality in
    place (task_dispatcher.cc:190-220); OnExpirationTimer drops servants whose lease ran out
    and keeps the order of the rest (:503-516)."""
    for i in range(5):
        td.keep_servant_alive("10.0.0.%d:8335" % i, ["a", "b"][: 1 + i % 2], 8, 16, i,
                              expires_in_ms=1000 * (1 + i))
    j = td.dump_internals()
    assert [s["location"] for s in j["servants"]] == ["10.0.0.%d:8335" % i for i in range(5)]
    assert j["servants_up"] == 5 and j["capacity"] == 40 and j["running_tasks"] == 0
    td.keep_servant_alive("10.0.0.1:8335", ["c"], 0, 32, 7, reason=4, expires_in_ms=9000,
                          reported="192.168.1.1:8335")
    j = td.dump_internals()
    s1 = j["servants"][1]
    assert s1["observed_location"] == "10.0.0.1:8335" and s1["reported_location"] == "192.168.1.1:8335"
    assert s1["not_accepting_task_reason"] == "NOT_ACCEPTING_TASK_REASON_BEHIND_NAT"
    assert s1["environments"] == ["c"] and s1["num_processors"] == 32 and "max_tasks" not in s1
    td.clock_advance_ms(2500)  # leases are 1, 9 (renewed), 3, 4, 5 s: only servant 0 is gone
    td.on_expiration_timer()
    j = td.dump_internals()
    assert [s.get("location", s.get("observed_location")) for s in j["servants"]] == [
        "10.0.0.1:8335", "10.0.0.2:8335", "10.0.0.3:8335", "10.0.0.4:8335"]
    td.clock_advance_ms(2000)  # t = 4.5 s: leases of 3 s and 4 s are gone
    td.on_expiration_timer()
    assert [s.get("location", s.get("observed_location")) for s in td.dump_internals()["servants"]] == [
        "10.0.0.1:8335", "10.0.0.4:8335"]


def test_capacity_available_in_dump(td):
    """GetCapacityAvailable (task_dispatcher.cc:283-313) as shown by DumpInternals."""
    td.keep_servant_alive("10.0.0.1:1", ["a"], 7, 16, 12)                      # min(7, 16-12)
    td.keep_servant_alive("10.0.0.2:1", ["a"], 7, 16, 40)                      # overloaded
    td.keep_servant_alive("10.0.0.3:1", ["a"], 7, 16, 0, total_memory=64 << 30,
                          memory_available=1 << 30)                             # low memory
    td.keep_servant_alive("10.0.0.4:1", ["a"], 7, 16, 0, total_memory=0,
                          memory_available=1 << 30)                             # not reported
    caps = [s["capacity_available"] for s in td.dump_internals()["servants"]]
    assert caps == [4, 0, 0, 7]


def _both():
    return R.RefDispatcher(), D.GpuTaskDispatcher(device=-1)


@needs_ref
def test_unknown_ids_and_bookkeeper_match_reference():
    """Answers that do not depend on placement, reference class vs ours, call for call
    (task_dispatcher_test.cc:80-102 without the grants; running_task_bookkeeper_test.cc:24-42)."""
    ref, td = _both()
    for d in (ref, td):
        d.keep_servant_alive("127.0.0.1:1234", ["digest"], 10, 10, 0, memory_available=G50)
    for d in (ref, td):
        assert not d.keep_task_alive(12345678, 1000)
        d.free_task(777)  # unknown id: silently ignored (task_dispatcher.cc:176-180)
    # a servant nobody knows: everything it reports is unknown (:241-243)
    a = ref.notify_servant_running_tasks("1.2.3.4:5", [5, 6, 5])
    b = td.notify_servant_running_tasks("1.2.3.4:5", [5, 6, 5])
    assert a == b == [5, 6, 5]
    # a known servant reporting grants the scheduler never made: all returned, in order,
    # duplicates kept, and none of them enters GetRunningTasks
    a = ref.notify_servant_running_tasks("127.0.0.1:1234", [1000002, 1000003, 1000002])
    b = td.notify_servant_running_tasks("127.0.0.1:1234", [1000002, 1000003, 1000002])
    assert a == b == [1000002, 1000003, 1000002]
    assert ref.get_running_tasks() == td.get_running_tasks() == []
    ref.close()
    td.close()


@needs_ref
def test_servant_expiry_matches_reference():
    ref, td = _both()
    for d in (ref, td):
        d.keep_servant_alive("10.0.0.1:1", ["x"], 4, 8, 0, expires_in_ms=1000)
        d.keep_servant_alive("10.0.0.2:1", ["x"], 4, 8, 0, expires_in_ms=5000)
    R.clock_advance_ms(2000)
    td.clock_advance_ms(2000)
    R.fire_timers()
    td.on_expiration_timer()
    # the expired servant is unknown now: its report comes back whole
    assert ref.notify_servant_running_tasks("10.0.0.1:1", [1]) == [1]
    assert td.notify_servant_running_tasks("10.0.0.1:1", [1]) == [1]
    assert [s["location"] for s in td.dump_internals()["servants"]] == ["10.0.0.2:1"]
    ref.close()
    td.close()
