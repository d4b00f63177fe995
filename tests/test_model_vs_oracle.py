"""CPU replay of the device pipeline (tests/model, built from the SAME placement
headers the kernels use) against the oracle: bit-exact placement, exact
utilisation, exact post-batch running_tasks."""
import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from tests.model import modelbind as M


def _check(sv, tk, chunk, fp64=False):
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    got, gutil, grun, st = M.dispatch(sv, tk, chunk, fp64)
    assert np.array_equal(got, want)
    assert np.array_equal(gutil, wutil)  # same IEEE division on both sides => bit-equal
    assert np.array_equal(grun, wrun)
    return st


@pytest.mark.parametrize("name,kw", cases.SMALL_CASES, ids=[c[0] for c in cases.SMALL_CASES])
@pytest.mark.parametrize("chunk", [0, 64, 1000])
def test_model_matches_oracle(name, kw, chunk):
    sv, tk = cases.random_case(**kw)
    _check(sv, tk, chunk)
    _check(sv, tk, chunk, fp64=True)


@pytest.mark.parametrize("name,sv,tk", cases.handmade_cases(),
                         ids=[c[0] for c in cases.handmade_cases()])
def test_model_handmade(name, sv, tk):
    for chunk in (0, 2, 7):
        _check(sv, tk, chunk)


def test_speculation_converges_quickly():
    sv, tk = cases.random_case(seed=21, n_tasks=100_000, n_servants=2000, n_envs=4,
                               unknown_env_frac=0.001)
    st = _check(sv, tk, 1024)
    assert st.n_chunks == 98 and st.rounds <= 4 and st.chunk_sims <= 2 * st.n_chunks


def test_disjoint_partitions_converge_in_two_rounds():
    """Disjoint environment partitions (independent parts of the registry, host_tables.h): every
    part is consumed at the rate of its own requests, so level guesses counted per part are
    exact and the replays agree at once; the global level needed about one round per chunk."""
    sv, tk = cases.random_case(seed=22, n_tasks=60_000, n_servants=1500, n_envs=4,
                               disjoint_envs=True, self_frac=0.0)
    st = _check(sv, tk, 256)
    assert st.n_chunks == 235 and st.rounds <= 2
    # with requests from servant hosts (holes) a few more, but nowhere near a round per chunk
    sv, tk = cases.random_case(seed=23, n_tasks=60_000, n_servants=1500, n_envs=4,
                               disjoint_envs=True, self_frac=0.2)
    assert _check(sv, tk, 256).rounds <= 6


def test_closed_form_slot_count_below_a_key():
    """first_slot_not_below (the per-servant count behind the key windows of the multi-GPU
    path) against a walk over the servant's slots, 1M random servants / keys."""
    assert M.lib().model_check_first_slot(7, 1_000_000) == 0


def test_tiny_pool_with_huge_servants_and_own_host_traffic():
    """Five servants with tens of thousands of slots each, one servant per class, a tenth of the
    requests from the servants' own hosts: a request from a host whose servant IS its class
    must not walk that servant's whole slot list to find out that nothing else is there
    (ClassLists::cls_single) — this batch took minutes before and takes a fraction of a second."""
    import time
    sv, tk = cases.random_case(seed=2009, n_tasks=150_000, n_servants=5, n_envs=3, self_frac=0.1,
                               unknown_env_frac=0.01, initial_running=True)
    t0 = time.time()
    _check(sv, tk, 64)
    assert time.time() - t0 < 30
