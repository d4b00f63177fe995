"""A bounded run of the randomised differential (tests/tools/fuzz_parity.py) inside `pytest -m gpu`:
200 seeded shapes — 1..8 digests, shared hosts, oversubscription, initial running_tasks, unknown
digests, self requests, forced chunk and ring sizes, batches committed in two halves — through
the HIP path and the oracle, every placement, utilisation and running_tasks column compared
bit for bit. The same seeds every run; the open-ended version is the developer tool itself."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("first_seed", [1000, 52000])
def test_bounded_fuzz_against_the_oracle(first_seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"), "240",
                          str(first_seed), "100"], cwd=ROOT, capture_output=True, text=True, timeout=400)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert "fuzz: 100 cases, 0 mismatches" in out.stdout, tail


@pytest.mark.gpu
def test_no_cliff_where_every_chunk_boundary_carries_a_hole(monkeypatch):
    """The corner the randomised differential found (fuzz_parity seed 72157): three
    servants offering 140k slots, one class, a tenth of 120k requests from the servants' own hosts —
    every chunk's true start state has a hole no level guess predicts, and parallel repair advances
    one chunk per pass (554 passes, 1.2 s). After a dozen passes one wave walks the rest."""
    import time

    import numpy as np

    from oracle import oraclebind as O
    from tests import cases
    from yadcc_amd import binding, pack
    seed = 72157
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, n_tasks=int(rng.choice([1, 63, 64, 65, 700, 5000, 30000, 120000])),
              n_servants=int(rng.choice([1, 3, 40, 300, 1500, 5000])),
              n_envs=int(rng.integers(1, 3)) if rng.random() < 0.5 else int(rng.integers(1, 9)),
              self_frac=float(rng.choice([0.0, 0.1, 0.5])), unknown_env_frac=float(rng.choice([0.0, 0.01])),
              min_version_20_frac=float(rng.choice([0.0, 0.5, 1.0])))
    assert kw["n_tasks"] == 120000 and kw["n_servants"] == 3, kw  # (the shape the tool drew for this seed)
    kw["initial_running"] = True       # (... and its other draws: initial load, rings of 256 entries)
    monkeypatch.setenv("YDC_RING_TOTAL", "256")
    sv, tk = cases.random_case(**kw)
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    c = binding.Context(device=0)
    c.upload_servants(pack.to_abi_columns(sv))
    c.dispatch(tk)
    t0 = time.perf_counter()
    got, gutil, grun = c.dispatch(tk)
    dt = time.perf_counter() - t0
    st = c.stats()
    c.close()
    assert np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
    assert st["rounds"] <= 24, st
    assert dt < 0.1, (dt, st)
    print("seed 72157: %d rounds, %.1f ms" % (st["rounds"], dt * 1e3))


@pytest.mark.gpu
def test_runs_of_the_requestors_own_slots_one_batch(monkeypatch):
    """... and the worst of that family from the same run (seed 700907, chunks of 4096, one digest,
    a batch that oversubscribes the pool): 114 s per case in the tool before, 30 ms for the batch with the run search."""
    import time

    import numpy as np

    from oracle import oraclebind as O
    from tests import cases
    from yadcc_amd import binding, pack
    monkeypatch.setenv("YDC_CHUNK_SIZE", "4096")
    sv, tk = cases.random_case(seed=700907, n_tasks=120000, n_servants=3, n_envs=1, self_frac=0.5,
                               unknown_env_frac=0.01, min_version_20_frac=1.0)
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    c = binding.Context(device=0)
    try:
        c.upload_servants(pack.to_abi_columns(sv))
        t0 = time.perf_counter()
        got, gutil, grun = c.dispatch(tk)
        dt = time.perf_counter() - t0
    finally:
        c.close()
    assert np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
    assert dt < 6.0, dt
    print("seed 700907: %.3f s" % dt)


@pytest.mark.gpu
def test_runs_of_the_requestors_own_slots(monkeypatch):
    """Another corner the randomised differential found (seed 704899: three servants offering 180k
    slots, half of 120k requests from their own hosts, chunks of 1024): a request whose own
    servant's slots form a run of thousands at the head of a class list walked over them one memory
    round trip at a time — 830 us per request, 31 s for the batch. The wave now looks for the end of
    such a run together, 64 entries at a time (match_kernel.h: general step)."""
    import time

    import numpy as np

    from oracle import oraclebind as O
    from tests import cases
    from yadcc_amd import binding, pack
    monkeypatch.setenv("YDC_CHUNK_SIZE", "1024")
    monkeypatch.setenv("YDC_RING_TOTAL", "1024")
    sv, tk = cases.random_case(seed=704899, n_tasks=120000, n_servants=3, n_envs=3, self_frac=0.5,
                               unknown_env_frac=0.01, min_version_20_frac=1.0)
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    n = len(tk["env_id"])
    c = binding.Context(device=0)
    try:
        c.upload_servants(pack.to_abi_columns(sv))
        t0 = time.perf_counter()
        cut = 73267  # (the tool committed this batch in two halves)
        a, ua, _ = c.dispatch({k: v[:cut] for k, v in tk.items()}, commit=True)
        b, ub, grun = c.dispatch({k: v[cut:] for k, v in tk.items()}, commit=True)
        dt = time.perf_counter() - t0
    finally:
        c.close()
    got, gutil = np.concatenate([a, b]), np.concatenate([ua, ub])
    assert np.array_equal(got, want) and np.array_equal(grun, wrun) and np.array_equal(gutil, wutil)
    assert dt < 2.0, dt
    print("seed 704899: %.3f s for both halves" % dt)
