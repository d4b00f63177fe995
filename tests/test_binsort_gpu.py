"""The bin sort (yadcc_amd/csrc/bin_sort.h: the slot order in three launches instead of the radix
sort's seven) against the oracle, side by side with the radix sort it stands in for, and its
fallback when a bin does not fit LDS. Which sort placed the slots shows in
ydc_stats::radix_passes (0: the bin sort)."""
import os

import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from tests.test_gpu_parity import check
from yadcc_amd import binding, pack, streaming, synth

pytestmark = pytest.mark.gpu


def _context(binsort):
    """A context with the bin sort on (and every bin sort checked against a host sort of the
    staged records, YDC_BINSORT_VERIFY) or off. The switches are read at ydc_create."""
    want = {"YDC_BINSORT": "1" if binsort else "0", "YDC_BINSORT_VERIFY": "1" if binsort else "0"}
    old = {k: os.environ.get(k) for k in want}
    os.environ.update(want)
    try:
        return binding.Context(device=0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def ctx_bin():
    c = _context(True)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_radix():
    c = _context(False)
    yield c
    c.close()


@pytest.mark.parametrize("name,kw", cases.SMALL_CASES, ids=[c[0] for c in cases.SMALL_CASES])
def test_small_cases_both_sorts(ctx_bin, ctx_radix, name, kw):
    sv, tk = cases.random_case(**kw)
    st = check(ctx_bin, sv, tk, "scan")
    if st["n_slots"] and st["n_classes"] <= 256:
        assert st["radix_passes"] == 0, st
    st = check(ctx_radix, sv, tk, "scan")
    assert st["radix_passes"] >= 1 or st["n_slots"] == 0


@pytest.mark.parametrize("name,sv,tk", cases.handmade_cases(),
                         ids=[c[0] for c in cases.handmade_cases()])
def test_handmade_cases_both_sorts(ctx_bin, ctx_radix, name, sv, tk):
    check(ctx_bin, sv, tk, "scan")
    check(ctx_radix, sv, tk, "scan")


def test_cfg2_takes_the_bin_sort(ctx_bin, ctx_radix):
    sv, tk = synth.make_config("cfg2")
    st = check(ctx_bin, sv, tk)
    assert st["radix_passes"] == 0 and st["n_slots"] > 100_000
    st = check(ctx_radix, sv, tk)
    assert st["radix_passes"] == 2


def test_busy_pool_and_parts(ctx_bin):
    """Servants that already run tasks (slots start mid-way through the key space) and disjoint
    environment partitions (the part id rides above the key: bins never mix parts)."""
    sv, tk = cases.random_case(seed=81, n_tasks=60_000, n_servants=1500, n_envs=4,
                               disjoint_envs=True, self_frac=0.2, initial_running=True)
    st = check(ctx_bin, sv, tk)
    assert st["radix_passes"] == 0
    sv, tk = cases.random_case(seed=82, n_tasks=120_000, n_servants=2500, n_envs=6,
                               unknown_env_frac=0.01, self_frac=0.1, initial_running=True)
    st = check(ctx_bin, sv, tk)
    assert st["radix_passes"] == 0 and st["n_classes"] > 8


def test_many_classes_bin_sort(ctx_bin):
    """65 .. 256 classes: class bits beyond one ballot word of the partition's match."""
    sv, tk = cases.random_case(seed=35, n_tasks=20_000, n_servants=1500, n_envs=7,
                               unknown_env_frac=0.002)
    st = check(ctx_bin, sv, tk)
    assert 64 < st["n_classes"] <= 256 and st["radix_passes"] == 0


def _tied_pool(n):
    """n identical idle servants: every slot key occurs n times — one bin holds them all."""
    sv = synth.make_servants(n, n_tasks_hint=4 * n, n_envs=1, seed=3)
    sv["version"][:] = 20
    sv["num_processors"][:] = 16
    sv["max_tasks"][:] = 8
    sv["current_load"][:] = 0
    sv["running_tasks"][:] = 0
    sv["priority"][:] = 2
    sv["total_memory"][:] = 64 << 30
    sv["memory_available"][:] = 32 << 30
    return sv


def test_bin_overflow_repeats_the_batch_with_the_radix_sort():
    """More equal keys than a bin's LDS buffer holds: the batch is gated out on the device,
    repeated with the radix sort (exact), and the context stays with the radix sort until the
    registry changes structure."""
    c = _context(True)
    try:
        sv = _tied_pool(6000)
        tk = synth.make_tasks(30_000, sv, seed=5, self_frac=0.1)
        st = check(c, sv, tk)
        assert st["radix_passes"] >= 1  # the repeat
        got, _, _ = c.dispatch(tk, commit=True)  # (now planned with the radix sort)
        assert c.stats()["radix_passes"] >= 1
        want, _, wrun = O.dispatch(sv, tk, "sorted")
        assert np.array_equal(got, want) and np.array_equal(c.get_running(), wrun)
        # another registry: the bin sort is tried again
        sv2, tk2 = cases.random_case(seed=83, n_tasks=5000, n_servants=300, n_envs=2)
        st = check(c, sv2, tk2)
        assert st["radix_passes"] == 0
    finally:
        c.close()


def test_bin_overflow_inside_a_streaming_tick():
    """The captured step uses the bin sort; a tick whose bins overflow is placed eagerly with the
    radix sort (its registry deltas are already applied) and the step is captured again."""
    sv = _tied_pool(5000)
    es = streaming.EventStream(sv, 3000, 1000)
    c = _context(True)
    try:
        c.upload_servants(pack.to_abi_columns(sv))
        c.stream_begin(es.hb + 8, 1000, 3000)
        for t in range(4):
            who, rows, rel, tk = es.next_tick()
            want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
            got = c.stream_tick(who, rows, rel, tk)
            assert np.array_equal(got, want), t
            es.commit(got)
            assert np.array_equal(c.get_running(), wrun), t
        c.stream_end()
    finally:
        c.close()
