"""Streaming mode (BASELINE.json configs[4]) against the oracle: every tick of a seeded event
stream — heartbeats, frees, a request batch — must place exactly like N sequential
WaitForStartingNewTask calls on the registry state the reference would have at that tick."""
import numpy as np
import pytest

from oracle import oraclebind as O
from yadcc_amd import binding, pack, streaming, synth

pytestmark = pytest.mark.gpu


def run_stream(n_servants, tasks_per_tick, frees_per_tick, ticks, n_envs=1, varying=False,
               capacity=None):
    sv = synth.make_servants(n_servants, n_tasks_hint=tasks_per_tick * 6, n_envs=n_envs, seed=42)
    es = streaming.EventStream(sv, tasks_per_tick, frees_per_tick, n_envs=n_envs)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    cap = capacity or tasks_per_tick
    ctx.stream_begin(es.hb + 8, max(frees_per_tick, 1), cap)
    for t in range(ticks):
        who, rows, rel, tk = es.next_tick()
        if varying and t % 3 == 1:  # fewer requests than the captured capacity
            tk = {k: v[: len(v) // 3] for k, v in tk.items()}
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
        got = ctx.stream_tick(who, rows, rel, tk)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "tick %d: first mismatch at request %d (gpu %d oracle %d), %d total" % (
            t, bad[0], got[bad[0]], want[bad[0]], bad.size)
        es.commit(got)
        assert np.array_equal(ctx.get_running(), wrun), "tick %d: running_tasks differ" % t
        assert ctx.stats()["granted"] == int((want < O.IDX_ENV_NOT_FOUND).sum())
    ctx.stream_end()
    ctx.close()
    return es


def test_stream_cfg5_shape():
    """10k requests/tick x 2k servants, 10k frees/tick, single compiler env."""
    es = run_stream(2000, 10_000, 10_000, ticks=12)
    assert es.tick_no == 12 and len(es.live) > 0


def test_stream_multi_env_varying_counts():
    run_stream(600, 3000, 2500, ticks=15, n_envs=3, varying=True)


def test_stream_saturated_pool_times_out():
    """More requests than the pool can ever hold: later ticks must time out exactly where the
    reference would (running_tasks at the cap, frees reopen slots)."""
    run_stream(150, 4000, 500, ticks=8, n_envs=2)


def test_stream_structural_heartbeat_recaptures():
    """A heartbeat that changes a servant's environments goes through the eager path and the
    step is captured again; results stay exact."""
    sv = synth.make_servants(300, n_tasks_hint=6000, n_envs=2, seed=5)
    es = streaming.EventStream(sv, 1000, 800, n_envs=2)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 800, 1000)
    for t in range(6):
        who, rows, rel, tk = es.next_tick()
        if t == 3:
            # servant who[0] drops / gains a compiler: structural change
            es.sv["env_mask"][who[0]] ^= np.uint64(3)
            es.abi = pack.to_abi_columns(es.sv)
            rows["env_mask"][0] = es.sv["env_mask"][who[0]]
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
        got = ctx.stream_tick(who, rows, rel, tk)
        assert np.array_equal(got, want), t
        es.commit(got)
        assert np.array_equal(ctx.get_running(), wrun)
    ctx.stream_end()
    ctx.close()
