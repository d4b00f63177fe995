"""The HOST class GpuTaskDispatcher without a GPU: gpu_task_dispatcher.cc + td_api.cc linked
against the CPU stand-in of the device API (tests/native/ydc_stub.cc over the CPU model of
tests/model) — locks, request combining, leases, digest / host interning, registry deltas and
servant expiry checked against the reference class itself on this box, plus the sanitizer
builds (`make -C tests/native tsan asan`). Placement arithmetic on the DEVICE
is what the -m gpu twin of this file (tests/test_task_dispatcher_gpu.py) checks; the product
library has no CPU placement."""
import ctypes as C
import os
import subprocess

import pytest

from oracle import refbind as R
from tests import td_scenarios as S
from yadcc_amd import dispatcher as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
_stub = None


class StubTaskDispatcher(D.GpuTaskDispatcher):
    @staticmethod
    def _load():
        global _stub
        if _stub is None:
            subprocess.check_call(["make", "-s", "-C", NATIVE, "all"])
            _stub = D.type_td_functions(C.CDLL(os.path.join(NATIVE, "libtd_stub.so")))
        return _stub


def make(**kw):
    d = StubTaskDispatcher(device=0, **kw)
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
@pytest.mark.parametrize("seed", [1, 2])
def test_event_stream_matches_reference(seed):
    S.event_stream_matches_reference(make, seed)


@needs_ref
def test_event_stream_150_digests():
    dump = S.event_stream_matches_reference(make, 11, n_digests=150, n_pool=120, steps=500)
    assert dump["gpu"]["environment_mask_words"] >= 2


@pytest.mark.parametrize("san", ["tsan", "asan"])
def test_sanitizer_build(san):
    """ThreadSanitizer / AddressSanitizer + UBSan builds of the host class under concurrent
    callers (tests/native/td_concurrency_test.cc)."""
    out = subprocess.run(["make", "-s", "-C", NATIVE, san], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "TD-CONCURRENCY-OK" in out.stdout, (out.stdout[-2000:],
                                                                     out.stderr[-4000:])


def test_flat_string_map_against_unordered_map():
    """The host class's string tables (yadcc_amd/csrc/flat_string_map.h: open addressing, backward
    shift on erase, miss filter) against std::unordered_map under random inserts / lookups /
    erases, ASan + UBSan build (tests/native/flat_map_test.cc)."""
    out = subprocess.run(["make", "-s", "-C", NATIVE, "flatmap"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "FLAT-MAP-OK" in out.stdout, (out.stdout[-2000:],
                                                                 out.stderr[-4000:])


@needs_ref
@pytest.mark.parametrize("seed", [1, 2])
def test_concurrent_callers_linearize(seed):
    """12 caller threads + a freer, a heartbeat thread and the clock / timer thread on 2000
    servants; the order in which their calls took effect, replayed one call at a time through the
    reference, gives the same answers and the same final state."""
    r = S.concurrent_callers_linearize(make, seed=seed)
    assert r["records"] > 5000 and r["kinds"].get("wait:0", 0) > 1000 and r["kinds"].get("timer", 0) > 3


@needs_ref
@pytest.mark.parametrize("shape", ["roomy", "saturated", "crowd"])
def test_native_callers_linearize(shape, tmp_path):
    """The same proof with the threads in C++ (tests/native/td_linearize.cc, through the C-ABI): 12
    callers at full speed, so that FreeTask queued behind somebody's device turn and the same
    thread's next request do meet in one turn (a build that applies queued frees after placing
    fails here on program order). roomy: nobody parks, frees are queued; saturated: most requests
    park and are retried at FreeTask's wake-ups."""
    subprocess.check_call(["make", "-s", "-C", NATIVE, "linearize"])
    args = {"roomy": dict(n_servants=300, calls=1500), "saturated": dict(n_servants=40, calls=3000, cap=1),
            # crowd: more parked requests than one device turn takes — placed again in segments, those
            # whose (digest, version, host) already came back Timeout in the turn not sent at all
            "crowd": dict(n_servants=40, n_threads=160, calls=200, cap=1)}[shape]
    r = S.native_linearize(os.path.join(NATIVE, "td_linearize_stub"), str(tmp_path / "lin.json"), seed=3, **args)
    assert r["records"] > 20000 and r["kinds"]["wait:0"] > (500 if shape == "crowd" else 3000)
    if shape != "roomy":
        assert r["retried_attempts"] > 5000 and r["kinds"]["wait:2"] > 10000


@needs_ref
def test_native_callers_linearize_tsan(tmp_path):
    """... and under ThreadSanitizer: no report, and the logged order still equals the reference's."""
    subprocess.check_call(["make", "-s", "-C", NATIVE, "linearize"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = S.native_linearize(os.path.join(NATIVE, "td_linearize_tsan"), str(tmp_path / "lin.json"), n_servants=60,
                           n_threads=8, calls=600, seed=4, env=env)
    assert r["records"] > 3000


@needs_ref
def test_event_stream_at_scale_small():
    """The scaled differential's event mix (bulk frees, batches of 1..256, structural heartbeats,
    expiries under thousands of live leases) on the stand-in, reduced: the host class's side of it."""
    r = S.event_stream_at_scale(make, seed=3, n_pool=300, prefill=5000, steps=2500)
    assert r["live_low_water"] >= 4500
