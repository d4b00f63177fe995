"""Streaming mode (BASELINE.json configs[4]) against the oracle: every tick of a seeded event
stream — heartbeats, frees, a request batch — must place exactly like N sequential
WaitForStartingNewTask calls on the registry state the reference would have at that tick."""
import numpy as np
import pytest

from oracle import oraclebind as O
from yadcc_amd import binding, pack, streaming, synth

pytestmark = pytest.mark.gpu


def run_stream(n_servants, tasks_per_tick, frees_per_tick, ticks, n_envs=1, varying=False,
               capacity=None, in_place=False):
    sv = synth.make_servants(n_servants, n_tasks_hint=tasks_per_tick * 6, n_envs=n_envs, seed=42)
    es = streaming.EventStream(sv, tasks_per_tick, frees_per_tick, n_envs=n_envs)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    cap = capacity or tasks_per_tick
    ctx.stream_begin(es.hb + 8, max(frees_per_tick, 1), cap)
    views = ctx.stream_buffers(es.hb + 8, max(frees_per_tick, 1), cap) if in_place else None
    for t in range(ticks):
        who, rows, rel, tk = es.next_tick()
        if varying and t % 3 == 1:  # fewer requests than the captured capacity
            tk = {k: v[: len(v) // 3] for k, v in tk.items()}
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
        if in_place:  # the tick assembled in the library's page-locked arena: nothing is copied
            views["upd_idx"][:len(who)] = who
            views["upd_rows"][:len(who)] = np.asarray(rows, dtype=binding.ROW_DTYPE)
            views["release_idx"][:len(rel)] = rel
            for k in ("env_id", "min_version", "requestor_ip"):
                views[k][:len(tk[k])] = tk[k]
            got = ctx.stream_tick_inplace(len(who), len(rel), len(tk["env_id"])).copy()
        else:
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


@pytest.mark.parametrize("zero_copy", ["1", "0"])
def test_stream_multi_env_varying_counts(zero_copy, monkeypatch):
    """zero_copy = 1: the captured step reads the tick's page-locked arena in place and k_finalize
    stores placement and outcome to page-locked memory (no copy node); 0: three copy nodes."""
    monkeypatch.setenv("YDC_STREAM_ZERO_COPY", zero_copy)
    monkeypatch.setenv("YDC_OUTCOME_STORE", zero_copy)
    run_stream(600, 3000, 2500, ticks=15, n_envs=3, varying=True)


def test_stream_assembled_in_place():
    """ydc_stream_buffers_get: the caller fills the page-locked arena itself and passes its pointers."""
    run_stream(800, 4000, 3000, ticks=12, n_envs=2, varying=True, in_place=True)


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


def test_stream_hosts_with_several_servants():
    """Registries where a host runs several servants take the sequential path on the device;
    inside the captured step that path is selected by a device-side flag."""
    sv = synth.make_servants(120, n_tasks_hint=3000, n_envs=2, seed=9, shared_ip_frac=0.3)
    es = streaming.EventStream(sv, 600, 400, n_envs=2)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 400, 600)
    for t in range(5):
        who, rows, rel, tk = es.next_tick()
        # a third of the requests come from servant hosts (some of them shared)
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "scan")
        got = ctx.stream_tick(who, rows, rel, tk)
        assert np.array_equal(got, want), t
        es.commit(got)
        assert np.array_equal(ctx.get_running(), wrun)
    ctx.stream_end()
    ctx.close()


def test_stream_new_servant_appends():
    """A heartbeat of an unknown servant (index == current count) appends it; the step is
    captured again and the newcomer takes requests from then on."""
    sv = synth.make_servants(40, n_tasks_hint=900, seed=12)
    es = streaming.EventStream(sv, 300, 100)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(64, 100, 300)
    who, rows, rel, tk = es.next_tick()
    es.commit(ctx.stream_tick(who, rows, rel, tk))
    # a big idle dedicated servant joins
    new = {"version": 20, "num_processors": 256, "current_load": 0, "max_tasks": 243,
           "priority": 1, "total_memory": 256 << 30, "memory_available": 128 << 30, "env_mask": 1,
           "ip": (10 << 24) + 9999, "port": 8335, "running_tasks": 0}
    for k, v in new.items():
        es.sv[k] = np.append(es.sv[k], np.array([v], dtype=es.sv[k].dtype))
    es.n += 1
    es.foreign = np.append(es.foreign, 0)
    es.running = np.append(es.running, 0)
    es.abi = pack.to_abi_columns(es.sv)
    row = np.zeros(1, dtype=binding.ROW_DTYPE)
    for k in ("version", "num_processors", "current_load", "max_tasks"):
        row[k] = es.sv[k][-1]
    row["flags"], row["ip_id"], row["env_mask"] = es.abi["flags"][-1], es.abi["ip_id"][-1], 1
    tk = synth.make_tasks(300, es.sv, seed=77)
    want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "scan")
    got = ctx.stream_tick(np.array([es.n - 1], np.uint32), row, np.empty(0, np.uint32), tk)
    assert np.array_equal(got, want) and (got == es.n - 1).sum() > 50
    ctx.n_servants = es.n
    assert np.array_equal(ctx.get_running(), wrun)
    ctx.stream_end()
    ctx.close()


def test_stream_wide_masks_env_change_and_append_inside_a_tick():
    """A registry with 150 digests (three mask words): ydc_stream_tick_wide carries the
    heartbeats' environment sets, so a servant may change what it advertises — or join — inside
    a tick (KeepServantAlive replaces the personality wholesale, task_dispatcher.cc:195-210).
    The narrow tick on such a table keeps a known servant's environments and refuses a new one."""
    n_envs = 150
    sv = synth.make_servants(200, n_tasks_hint=4000, n_envs=n_envs, seed=21)
    es = streaming.EventStream(sv, 800, 500, n_envs=n_envs)
    assert es.abi["env_mask"].ndim == 2 and es.abi["env_mask"].shape[1] == 3
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 500, 800)
    for t in range(7):
        who, rows, rel, tk = es.next_tick()
        if t == 2:
            # who[0] drops every digest of its second mask word and gains digest 5
            es.sv["env_mask"][who[0], 1] = 0
            es.sv["env_mask"][who[0], 0] |= np.uint64(1 << 5)
            es.abi = pack.to_abi_columns(es.sv)
        if t == 4:
            # a new servant that advertises digests 3, 70 and 140 joins in this tick
            for k in es.sv:
                add = es.sv[k][:1].copy()
                es.sv[k] = np.concatenate([es.sv[k], add])
            s = es.n
            es.sv["env_mask"][s] = 0
            for d in (3, 70, 140):
                es.sv["env_mask"][s, d // 64] |= np.uint64(1) << np.uint64(d % 64)
            es.sv["ip"][s] = (10 << 24) + 77777
            es.sv["num_processors"][s], es.sv["max_tasks"][s] = 128, 120
            es.sv["current_load"][s], es.sv["running_tasks"][s], es.sv["priority"][s] = 0, 0, 1
            es.n += 1
            es.foreign = np.append(es.foreign, 0)
            es.running = np.append(es.running, 0)
            es.abi = pack.to_abi_columns(es.sv)
            who = np.append(who, np.uint32(s)).astype(np.uint32)
            rows = np.concatenate([rows, np.zeros(1, dtype=binding.ROW_DTYPE)])
            for k in ("version", "num_processors", "current_load", "max_tasks"):
                rows[k][-1] = es.sv[k][s]
            rows["flags"][-1], rows["ip_id"][-1] = es.abi["flags"][s], es.abi["ip_id"][s]
            tk = synth.make_tasks(800, es.sv, n_envs=n_envs, seed=5000)
        masks = es.abi["env_mask"][who]
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
        if t == 4:
            with pytest.raises(binding.YdcError, match="ydc_stream_tick_wide"):
                ctx.stream_tick(who, rows, rel, tk)  # narrow rows cannot introduce a servant here
            # ... nor behind a structural heartbeat of a known servant (the scan for structure
            # stops there; the newcomer must still be seen and refused, not read out of bounds)
            rows2 = rows.copy()
            rows2["version"][0] += 1
            with pytest.raises(binding.YdcError, match="ydc_stream_tick_wide"):
                ctx.stream_tick(who, rows2, rel, tk)
        got = ctx.stream_tick(who, rows, rel, tk, env_masks=masks)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (t, bad[:5], got[bad[:5]], want[bad[:5]])
        es.commit(got)
        assert np.array_equal(ctx.get_running(), wrun), t
        if t == 4:
            assert (got == es.n - 1).sum() > 0  # the newcomer takes requests at once
    ctx.stream_end()
    ctx.close()


def test_stream_cfg5_200_ticks_against_the_reference():
    """BASELINE.json configs[4], 200 ticks (10k requests + 10k frees + 200 heartbeats each) through
    the captured step, tick by tick against what the VERBATIM reference answered on the same
    stream (tests/golden/ref_cfg5_stream_200_ticks.npz, generator
    tests/golden/make_stream_golden.py: digests of every tick's placement and running_tasks)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_cfg5_stream_200_ticks.npz"))
    sv, _ = synth.make_config("cfg5")
    es = streaming.EventStream(sv, 10_000, 10_000)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 10_000, 10_000)
    for t in range(int(fx["ticks"])):
        who, rows, rel, tk = es.next_tick()
        got = ctx.stream_tick(who, rows, rel, tk)
        assert synth.placement_hash(got) == int(fx["digest"][t]), "tick %d: placement differs" % t
        assert int((got < binding.IDX_ENV_NOT_FOUND).sum()) == int(fx["granted"][t])
        es.commit(got)
        if t % 10 == 9 or t < 3:
            assert synth.placement_hash(ctx.get_running()) == int(fx["run_digest"][t]), t
    ctx.stream_end()
    ctx.close()


def test_stream_more_than_256_classes_runs_eagerly():
    """A registry with ~600 servant classes (150 digests, individual compiler sets): the step
    cannot be captured (the many-class path has host-checked rounds), so every tick is enqueued
    instead of replayed — same interface, same placement as the reference order."""
    n_envs = 150
    sv = synth.make_servants(700, n_tasks_hint=9000, n_envs=n_envs, seed=23)
    es = streaming.EventStream(sv, 1500, 1000, n_envs=n_envs)
    ctx = binding.Context(device=0)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 1000, 1500)
    for t in range(5):
        who, rows, rel, tk = es.next_tick()
        want, _, wrun = O.dispatch(es.registry_snapshot(), tk, "sorted")
        got = ctx.stream_tick(who, rows, rel, tk, env_masks=es.abi["env_mask"][who])
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (t, bad[:5], got[bad[:5]], want[bad[:5]])
        es.commit(got)
        assert np.array_equal(ctx.get_running(), wrun), t
        st = ctx.stats()
        assert st["n_classes"] > 256 and st["granted"] == int((want < O.IDX_ENV_NOT_FOUND).sum())
    ctx.stream_end()
    ctx.close()
