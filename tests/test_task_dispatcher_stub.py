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
