"""GpuTaskDispatcher on the MI355X (libydc.so) against the reference class itself: the
scenarios of tests/td_scenarios.py — the reference's own gtest cases call for call, wake-up
order, and seeded random event streams (3 digests; 150 digests = multi-word environment
masks) compared answer for answer with the reference's translation units compiled verbatim."""
import pytest

from oracle import refbind as R
from tests import td_scenarios as S
from yadcc_amd import dispatcher as D

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")


def make(**kw):
    d = D.GpuTaskDispatcher(device=0, **kw)
    assert d.device_status == 0
    return d


@pytest.fixture
def td():
    d = make()
    yield d
    d.close()


def test_golden_all(td):
    S.golden_all(td)


def test_golden_prefer_dedicated(td):
    S.golden_prefer_dedicated(td)


def test_golden_load_balance(td):
    S.golden_load_balance(td)


def test_lease_table_churn():
    S.lease_table_churn(make)


def test_blocking_wait_is_woken_by_free_task():
    S.blocking_wait_is_woken_by_free_task(make)


def test_heartbeat_wakes_nobody():
    S.heartbeat_wakes_nobody(make)


def test_timer_tick_wakes_parked_waiters():
    S.timer_tick_wakes_parked_waiters(make)


def test_concurrent_callers_are_combined():
    S.concurrent_callers_are_combined(make)


def test_location_that_does_not_fit_is_an_error():
    S.location_that_does_not_fit_is_an_error(make)


def test_address_forms():
    S.address_forms(make)


@needs_ref
@pytest.mark.parametrize("seed", [5, 6])
def test_address_prefix_forms(seed):
    S.address_prefix_forms(make, seed)


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_event_stream_matches_reference(seed):
    S.event_stream_matches_reference(make, seed)


@needs_ref
@pytest.mark.parametrize("seed", [11, 12])
def test_event_stream_150_digests(seed):
    """More than 64 distinct live compiler digests (the reference has no limit,
    task_dispatcher.h:93-94, .cc:55-63): multi-word environment masks on the device."""
    dump = S.event_stream_matches_reference(make, seed, n_digests=150, n_pool=120, steps=500)
    assert dump["gpu"]["environment_mask_words"] >= 2


def test_native_scheduler_harness():
    """tests/native/harness_test: the dispatcher driven natively (C++) through
    SchedulerHarness, the call pattern of the reference's SchedulerServiceImpl."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "harness_test")
    # (always through make: a binary built against an older gpu_task_dispatcher.h must not survive)
    subprocess.check_call(["make", "-s", "tests/native/harness_test"], cwd=root)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "HARNESS-OK" in out.stdout, (out.stdout, out.stderr)


@needs_ref
@pytest.mark.parametrize("shape", ["roomy", "saturated"])
def test_concurrent_callers_linearize(shape):
    """12 caller threads + freer + heartbeats + clock / timer on the real library: the order in
    which the calls took effect (ydc_td_oplog_*), replayed one call at a time through the
    reference class, gives the same (status, task id, location) for every call and the same final
    DumpInternals; every thread's own calls appear in program order and no call overtakes one
    that had returned before it was made (tests/td_scenarios.py:verify_linearizable)."""
    kw = dict(n_servants=2000, calls=500) if shape == "roomy" else dict(n_servants=40, calls=1500, max_tasks_cap=1)
    r = S.concurrent_callers_linearize(make, seed=7, **kw)
    assert r["records"] > 5000 and r["kinds"].get("wait:0", 0) > 500
    if shape == "saturated":
        assert r["retried_attempts"] > 500


@needs_ref
@pytest.mark.parametrize("shape", ["roomy", "saturated", "crowd"])
def test_native_callers_linearize(shape, tmp_path):
    """... with the threads in C++ (tests/native/td_linearize.cc linked against libydc.so): 16
    callers at full speed on 2000 servants (roomy: queued FreeTasks meet the same thread's next
    request inside one device turn) and on a saturated pool (parked waiters retried at wake-ups)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "tests/native/td_linearize_gpu"], cwd=root)
    args = {"roomy": dict(n_servants=2000, n_threads=16, calls=2500),
            "saturated": dict(n_servants=40, n_threads=16, calls=3000, cap=1),
            # more parked requests than a device turn takes: placed again in segments (gpu_task_dispatcher.cc:
            # UnsafeDispatchSegmented), the known Timeouts not sent at all
            "crowd": dict(n_servants=40, n_threads=200, calls=300, cap=1)}[shape]
    r = S.native_linearize(os.path.join(root, "tests", "native", "td_linearize_gpu"), str(tmp_path / "lin.json"),
                           seed=5, **args)
    assert r["records"] > 40000 and r["kinds"]["wait:0"] > (800 if shape == "crowd" else 5000)
    if shape == "roomy":
        assert r["requests_per_device_turn"] > 1.2  # callers were combined
    else:
        assert r["retried_attempts"] > 5000


@needs_ref
def test_event_stream_at_scale():
    """2000 servants, >= 5*10^4 live leases throughout, 2*10^4 events with grant batches of 1..256:
    the resident tick kernel, the launched tick and the batch pipeline all serve calls of ONE
    stream, between structural heartbeats, bulk frees and timer ticks — every answer and the final
    DumpInternals equal to the reference class's (~1 min, most of it the reference)."""
    r = S.event_stream_at_scale(make, seed=21)
    assert r["live_low_water"] >= 45_000 and r["servants_at_end"] > 1500
    g = r["gpu"]
    assert g["tick_resident_calls"] > 500 and g["tick_launched_calls"] > 50 and g["pipeline_batches"] > 500, g
