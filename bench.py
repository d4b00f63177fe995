#!/usr/bin/env python3
"""bench.py — task->servant assignments/s of the MI355X dispatch path.

One "step" = one pass of the hot path over one batch: BASELINE.json configs[1]
(100k pending requests x 2k servants, single compiler env) dispatched against the
resident servant table, request columns and result buffers already in HBM. Each step
starts from the same snapshot (no COMMIT), so every step does identical work.

`value` is the HBM-resident rate (the bench contract: inputs resident when the timed
region starts). The metric as SURVEY.md §8(d) words it — batch visible to the dispatcher ->
all results visible to the host, i.e. H2D + kernels + D2H through ydc_dispatch — is in
`end_to_end` on the same line (rate, p50 / p99 over >= 100 batches); `value_definition` says
which is which, and `value_synchronous` / `value_pipelined` / `value_end_to_end` carry the
three rates side by side.

N = 1 (default cfg2): the line also carries `configs` — compact records of BASELINE.json
configs[2] (cfg3), configs[3]'s batch on one GPU (cfg4) and configs[4] (cfg5, streaming), each
timed in this very run and pinned to the committed fixtures of the verbatim reference
(tests/golden: first 50k placements / 200 ticks) — no CPU replay, a few seconds.

N > 1: `python bench.py --gpus N` launches its N ranks itself when no launcher did
(WORLD_SIZE unset): rank r takes device r % (visible devices), so the command also runs on
a box with fewer GPUs than ranks (the ranks then share a device over the mailbox transport).
Under torch.distributed.run (one rank per GPU) RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come
from the environment; `--gpus` must agree with WORLD_SIZE. torch.distributed (gloo) is used
for the rendezvous, the barriers and the max-over-ranks of the wall time; the data path of
the sharded batch is RCCL inside libydc.so. `--scaling weak` (default): G ranks place ONE
global batch of G x (the config's requests) on a pool of G x (the config's servants);
`--scaling strong`: the config's batch itself is cut into G rank ranges. For N > 1 the line
also carries `strong_cfg4`: BASELINE.json configs[3] as specified (4M requests x 16k
servants cut over the N ranks), timed in the same run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "task-to-servant assignments/sec on synthetic pool"


def percentile(a, q):
    a = np.sort(np.asarray(a))
    return float(a[min(len(a) - 1, int(np.ceil(q * len(a))) - 1)]) if len(a) else 0.0


def cpu_baseline(sv, tk, max_tasks=None):
    """The reference's own TaskDispatcher (oracle/_ref, compiled verbatim) on this host,
    1 thread (everything in the reference runs under one lock), same snapshot."""
    from oracle import oraclebind as O
    from oracle import refbind as R
    if max_tasks and len(tk["env_id"]) > max_tasks:
        tk = {k: v[:max_tasks] for k, v in tk.items()}
    if R.available():
        d = R.RefDispatcher()
        d.load_servants(sv)
        idx, _, secs, lat = d.dispatch_batch(tk, want_latency=True)
        d.close()
        granted = int((idx < R.IDX_ENV_NOT_FOUND).sum())
        # A fairer "optimised CPU" bar: the SoA restatement in slot order
        # (oracle_dispatch_sorted, one thread), same requests.
        p0 = time.perf_counter()
        pidx, _, _ = O.dispatch(sv, tk, "sorted")
        psecs = time.perf_counter() - p0
        return idx, {
            "value": granted / secs, "unit": "assignments/s", "cores": 1, "kind": "reference",
            "sample": "first %d requests of the batch: sequential WaitForStartingNewTask calls, "
                      "%d servants, %.2f s" % (len(idx), len(sv["version"]), secs),
            "p99_latency_us": percentile(lat, 0.99) / 1e3,
            "host_cores_available": os.cpu_count(),
            "soa_port_value": int((pidx < O.IDX_ENV_NOT_FOUND).sum()) / psecs,
            "soa_port_sample": "oracle_dispatch_sorted (SoA, slot-order formulation), 1 thread, "
                               "same requests, %.3f s; identical placement: %s" % (
                                   psecs, bool(np.array_equal(pidx, idx))),
        }
    t0 = time.perf_counter()
    idx, _, _ = O.dispatch(sv, tk, "scan")
    secs = time.perf_counter() - t0
    granted = int((idx < O.IDX_ENV_NOT_FOUND).sum())
    return idx, {"value": granted / secs, "unit": "assignments/s", "cores": 1, "kind": "port",
                 "sample": "first %d requests through oracle_dispatch_scan, %.2f s" % (
                     len(idx), secs),
                 "host_cores_available": os.cpu_count()}


def golden(name):
    """A committed fixture of the verbatim reference (tests/golden/, generators beside them)."""
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def fixture_prefix(cfg, sv, tk):
    """The verbatim reference's placement of the first requests of `cfg` (tests/golden/
    ref_<cfg>_prefix_50k.npz, tests/golden/make_golden.py), or None when (sv, tk) are not the
    inputs the fixture was generated from (another digest count, shared hosts, ...)."""
    import hashlib
    try:
        z = golden("ref_%s_prefix_50k.npz" % cfg)
    except OSError:
        return None
    n = int(z["prefix"])
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    return z["ref_servant_idx"] if h.hexdigest() == str(z["input_sha256"]) else None


def fixture_digests_ok(cfg, sv, tk, servant_idx):
    """The long prefixes pinned block by block to the verbatim reference (tests/golden/
    ref_<cfg>_prefix_digests.npz, tests/golden/make_prefix_digests.py: cfg3's first 400k requests —
    past the dedicated-tier boundary —, cfg4's first 200k). -> (all blocks equal, requests
    covered), or (None, 0) without a fixture for these inputs."""
    import hashlib
    from yadcc_amd import synth
    try:
        z = golden("ref_%s_prefix_digests.npz" % cfg)
    except OSError:
        return None, 0
    n, block = int(z["prefix"]), int(z["block"])
    h = hashlib.sha256()
    for k in sorted(sv):
        h.update(np.ascontiguousarray(sv[k]).tobytes())
    for k in sorted(tk):
        h.update(np.ascontiguousarray(tk[k][:n]).tobytes())
    if h.hexdigest() != str(z["input_sha256"]):
        return None, 0
    ok = all(synth.placement_hash(servant_idx[b * block:(b + 1) * block]) == int(z["digest"][b])
             for b in range(n // block))
    return bool(ok), n


def stream_record(args, steps, warmup, ref_ticks, device=0):
    """BASELINE.json configs[4]: 10k requests/tick x 2k servants with rolling heartbeats (10 % of
    the servants per tick) and 10k frees per tick; the whole tick is one replay of a captured
    hipGraph (ydc_stream_tick). A step is a tick; only the tick call is timed (the event
    generator is host-side test scaffolding). Host buffers in, host results out. Every one of
    the first 200 ticks is compared with what the verbatim reference answered on the same
    stream (tests/golden/ref_cfg5_stream_200_ticks.npz); with ref_ticks > 0 the reference class
    also replays the first ticks beside it (heartbeats, frees by grant id, sequential
    WaitForStartingNewTask calls): cpu_baseline + per-tick parity."""
    from yadcc_amd import binding, pack, streaming, synth
    sv, _ = synth.make_config("cfg5")
    es = streaming.EventStream(sv, 10_000, 10_000)
    ctx = binding.Context(device=device)
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 10_000, 10_000)
    try:
        fx = golden("ref_cfg5_stream_200_ticks.npz")
        fx_ticks = int(fx["ticks"])
    except OSError:
        fx, fx_ticks = None, 0
    ref = ref_ids = None
    ref_secs, ref_granted, ref_lat, parity = 0.0, 0, [], True
    fx_ok, fx_seen = True, 0
    if ref_ticks:
        from oracle import refbind as R
        if R.available():
            ref = R.RefDispatcher()
            ref.load_servants(sv)
            ref_ids = np.empty(0, np.uint64)  # grant id of every live grant, stream order
    lat, granted = [], 0
    for t in range(warmup + steps):
        who, rows, rel, tk = es.next_tick()
        if ref is not None and t < ref_ticks:
            # heartbeats: the same personalities, current_load as the stream reports it
            hb = {k: v[who] for k, v in es.sv.items()}
            hb["running_tasks"] = np.zeros(len(who), np.uint32)  # (kept by a renewal anyway)
            ref.load_servants(hb)
            ref.free_tasks(ref_ids[es.last_freed])
            ref_ids = ref_ids[es.last_kept]
            ridx, rids, secs, rl = ref.dispatch_batch(tk, want_latency=True)
            ref_secs += secs
            ref_lat.append(rl)
            ok = ridx < R.IDX_ENV_NOT_FOUND
            ref_granted += int(ok.sum())
            ref_ids = np.concatenate([ref_ids, rids[ok]])
        s0 = time.perf_counter()
        got = ctx.stream_tick(who, rows, rel, tk)
        dt = time.perf_counter() - s0
        if ref is not None and t < ref_ticks:
            parity &= bool(np.array_equal(got, ridx))
        if t < fx_ticks:
            fx_ok &= synth.placement_hash(got) == int(fx["digest"][t])
            fx_seen += 1
        es.commit(got)
        if t >= warmup:
            lat.append(dt)
            granted += int((got < binding.IDX_ENV_NOT_FOUND).sum())
    st = ctx.stats()
    # The same ticks assembled where the captured step reads them (ydc_stream_buffers_get): what
    # is left of a tick when nothing is copied on either side. Same stream, the next ticks.
    in_place = []
    views = ctx.stream_buffers(es.hb + 8, 10_000, 10_000)
    for t in range(min(300, max(50, steps))):
        who, rows, rel, tk = es.next_tick()
        views["upd_idx"][:len(who)] = who
        views["upd_rows"][:len(who)] = np.asarray(rows, dtype=binding.ROW_DTYPE)
        views["release_idx"][:len(rel)] = rel
        for k in ("env_id", "min_version", "requestor_ip"):
            views[k][:len(tk[k])] = tk[k]
        s0 = time.perf_counter()
        got = ctx.stream_tick_inplace(len(who), len(rel), len(tk["env_id"]))
        in_place.append(time.perf_counter() - s0)
        es.commit(got.copy())
    # The same ticks with the step enqueued kernel by kernel instead of replayed from its hipGraph
    # (YDC_TUNE=stream_graph=0; a context of its own, the stream from its start): the configuration
    # names the captured step, so that is what `value` is; this is what the capture costs or saves.
    eager = None
    try:
        os.environ["YDC_STREAM_GRAPH"] = "0"
        ctx2 = binding.Context(device=device)
    finally:
        os.environ.pop("YDC_STREAM_GRAPH", None)
    try:
        es2 = streaming.EventStream(sv, 10_000, 10_000)
        ctx2.upload_servants(pack.to_abi_columns(sv))
        ctx2.stream_begin(es2.hb + 8, 10_000, 10_000)
        el, e_ok, n2 = [], True, min(300, max(60, steps))
        for t in range(20 + n2):
            who, rows, rel, tk = es2.next_tick()
            s0 = time.perf_counter()
            got = ctx2.stream_tick(who, rows, rel, tk)
            dt = time.perf_counter() - s0
            if t < fx_ticks:
                e_ok &= synth.placement_hash(got) == int(fx["digest"][t])
            es2.commit(got)
            if t >= 20:
                el.append(dt)
        eager = {"ms_per_step": 1e3 * sum(el) / len(el), "p99_ms": 1e3 * percentile(el, 0.99), "ticks": len(el),
                 "parity_vs_reference_fixture": bool(e_ok) if fx_ticks else None,
                 "what": "the same step enqueued kernel by kernel (YDC_TUNE=stream_graph=0) instead of replayed "
                         "from the captured hipGraph"}
        ctx2.stream_end()
    finally:
        ctx2.close()
    out = {
        "metric": METRIC,
        "value": granted / sum(lat), "unit": "assignments/s", "n_gpus": 1,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * sum(lat) / len(lat),
        "tick_enqueued_eagerly": eager,
        "tick_assembled_in_place": {"ms_per_step": 1e3 * sum(in_place) / len(in_place),
                                    "p99_ms": 1e3 * percentile(in_place, 0.99), "ticks": len(in_place),
                                    "what": "the tick written into the library's page-locked arena by the caller "
                                            "(ydc_stream_buffers_get): no staging copy on either side"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "value_definition": "end to end per tick: host buffers in (one H2D copy), registry "
                            "deltas + batch + commit, results back on the host (PCIe included)",
        "config": {"workload": "cfg5 streaming: 10000 requests + 10000 frees + %d heartbeats per "
                               "tick x %d servants, hipGraph-captured step" % (es.hb, es.n),
                   "parallelism": "1 GPU", "inputs": "host buffers per tick (PCIe included)"},
        "p99_dispatch_latency_ms": 1e3 * percentile(lat, 0.99),
        "p50_dispatch_latency_ms": 1e3 * percentile(lat, 0.50),
        "latency_samples": len(lat),
        "stats": {k: v for k, v in st.items() if k != "stage_ms"},
        "parity_vs_reference_fixture": bool(fx_ok) if fx_seen else None,
        "fixture_ticks": fx_seen,
    }
    # 16 B per request + 40 B per servant per tick over the whole tick: the
    # captured step is one graph launch, so the "dominant kernel" is the step itself.
    alg = 16 * 10_000 + 40 * es.n
    tick_s = sum(lat) / len(lat)
    out["roofline"] = {"bound": "hbm", "kernel": "captured step (one hipGraph launch)",
                       "achieved": alg / tick_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": alg / tick_s / 8e12, "traffic": None,
                       "algorithmic_bytes_per_launch": alg, "avg_launch_us": tick_s * 1e6}
    if ref is not None:
        ref.close()
        rl = np.concatenate(ref_lat)
        out["cpu_baseline"] = {
            "value": ref_granted / ref_secs, "unit": "assignments/s", "cores": 1,
            "kind": "reference",
            "sample": "the first %d ticks of the same stream replayed through the reference class "
                      "(heartbeats, frees by grant id, %d sequential WaitForStartingNewTask calls, "
                      "%.2f s inside them)" % (ref_ticks, len(rl), ref_secs),
            "p99_latency_us": percentile(rl, 0.99) / 1e3,
            "ms_per_tick": 1e3 * ref_secs / ref_ticks,
            "host_cores_available": os.cpu_count()}
        out["parity_vs_cpu_baseline"] = parity
        out["parity_ticks"] = ref_ticks
    ctx.stream_end()
    ctx.close()
    return out


def td_surface(with_reference=True):
    """The preserved TaskDispatcher surface (ydc_td_*: strings in, grant ids and location strings
    out — what SchedulerServiceImpl would call) timed natively by tools/td_native_bench, one
    caller thread: requests/s through ydc_td_wait_for_starting_new_tasks in batches of 10k and
    100k, KeepServantAlive + NotifyServantRunningTasks per second and GetRunningTasks per second
    at 16k servants / 10^6 live leases and at 2k servants / 10^5 leases — the
    scale at which the verbatim reference is timed beside it (a bounded sample: its heartbeat is
    two scans over every lease, task_dispatcher.cc:222-277,453-476)."""
    tool = os.path.join(ROOT, "tools", "td_native_bench")
    if not os.path.exists(tool):
        return {"error": "tools/td_native_bench is not built (make native)"}
    out = {"unit": "calls/s, one caller thread", "tool": "tools/td_native_bench (C++, through the C-ABI)"}
    for name, argv in (("wait_batch_10k", ["wait", "2000", "10000", "50"]),
                       ("wait_batch_100k", ["wait", "2000", "100000", "20"]),
                       ("heartbeat_16k_servants_1M_leases", ["heartbeat", "16000", "1000000", "3"]),
                       ("heartbeat_2k_servants_100k_leases", ["heartbeat", "2000", "100000", "5"])):
        try:
            r = subprocess.run([tool] + argv, capture_output=True, text=True, timeout=300, cwd=ROOT)
            out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {
                "error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # The reference's real call shape: one WaitForStartingTask RPC = waiters + 1 grants
    # (daemon/local/task_grant_keeper.cc:145-146) — latency per CALL for 1 .. 256 requests, a
    # FreeTask of the grants between two calls, a heartbeat before every fourth; the one-launch /
    # resident path (ydc_dispatch_tick) and, for the crossover, the batch pipeline alone
    # (YDC_TUNE=small_batch=0) and the launch-per-call form (resident=0).
    out["latency"] = {"unit": "us per call, one caller thread; p50 / p99 over 1000 calls",
                      "what": "ydc_td_wait_for_starting_new_task (single_1) / ..._tasks (batch_n), registry warm, "
                              "grants freed between calls, a heartbeat before every fourth call"}
    for S in (2000, 16000):
        rec = {}
        for name, tune in (("tick", None), ("tick_launch_per_call", "resident=0"), ("batch_pipeline_only", "small_batch=0")):
            env = dict(os.environ)
            if tune:
                env["YDC_TUNE"] = ",".join(x for x in (tune, env.get("YDC_TUNE", "")) if x)
            try:
                r = subprocess.run([tool, "latency", str(S), "1000" if name == "tick" else "300"], capture_output=True,
                                   text=True, timeout=300, cwd=ROOT, env=env)
                rec[name] = json.loads(r.stdout.strip().splitlines()[-1])["per_call_us"] if r.returncode == 0 else {
                    "error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
            except Exception as e:  # noqa: BLE001
                rec[name] = {"error": str(e)}
        t, b = rec.get("tick", {}), rec.get("batch_pipeline_only", {})
        cross = [int(k.split("_")[1]) for k in t if k.startswith("batch_") and k in b and
                 isinstance(t[k], dict) and b[k]["p50"] < t[k]["p50"]]
        rec["crossover_batch"] = min(cross) if cross else None
        if with_reference:
            try:
                rec["reference"] = reference_latency(S)
            except Exception as e:  # noqa: BLE001
                rec["reference"] = {"error": str(e)}
        out["latency"]["servants_%d" % S] = rec
    # Many RPC handlers at once: 1 .. 32 caller threads, each a loop of one WaitForStartingNewTask +
    # FreeTask from its own requestor address. Whoever holds the dispatcher's lock places everything
    # that is queued in one device turn; the others spin for their answer. (The reference serialises
    # its callers on one lock around an 18 us scan: <= 5.5e4 calls/s whatever the thread count.)
    try:
        r = subprocess.run([tool, "concurrent", "2000", "2000"], capture_output=True, text=True, timeout=300, cwd=ROOT)
        out["concurrent_callers_2k_servants"] = json.loads(r.stdout.strip().splitlines()[-1])["threads"] if r.returncode == 0 else {
            "error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
    except Exception as e:  # noqa: BLE001
        out["concurrent_callers_2k_servants"] = {"error": str(e)}
    # The 1 s expiration timer running, as the reference always has it (task_dispatcher.cc:81-82,
    # 498-536): single-request latency with 10^5 / 10^6 live leases in the table, timer off and on.
    for name, argv in (("timer_2k_servants_100k_leases", ["timer", "2000", "100000", "3"]),
                       ("timer_16k_servants_1M_leases", ["timer", "16000", "1000000", "4"])):
        try:
            r = subprocess.run([tool] + argv, capture_output=True, text=True, timeout=300, cwd=ROOT)
            out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {
                "error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    # The reference's admitted scaling problem (task_dispatcher.h:281-288): K waiters parked on a
    # saturated pool, one FreeTask at a time. Same workload on both sides (tools/parked_workload.h).
    try:
        r = subprocess.run([tool, "parked", "200", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        out["parked_waiters"] = json.loads(r.stdout.strip().splitlines()[-1])["waiters"] if r.returncode == 0 else {
            "error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
    except Exception as e:  # noqa: BLE001
        out["parked_waiters"] = {"error": str(e)}
    if with_reference:
        try:
            out["parked_waiters_reference"] = reference_parked()
        except Exception as e:  # noqa: BLE001
            out["parked_waiters_reference"] = {"error": str(e)}
    if with_reference:
        try:
            out["reference"] = reference_td_surface()
        except Exception as e:  # noqa: BLE001
            out["reference"] = {"error": str(e)}
        ref, ours = out["reference"], out.get("heartbeat_2k_servants_100k_leases", {})
        if "heartbeats_per_s" in ref and "heartbeats_per_s" in ours:
            out["heartbeat_speedup_vs_reference_same_scale"] = ours["heartbeats_per_s"] / ref["heartbeats_per_s"]
            # (per listed entry: the reference's sample only has the sampled servants' reports)
            out["get_running_tasks_entries_speedup_vs_reference"] = (
                ours["get_running_tasks_per_s"] * ours["running_tasks_listed"] /
                (ref["get_running_tasks_per_s"] * max(ref["running_tasks_listed"], 1)))
    return out


def reference_latency(n_servants, calls=None):
    """Per-call latency of the verbatim reference's WaitForStartingNewTask (oracle/_ref, one
    thread) on the pool of tools/td_native_bench; a batch of n = n consecutive calls, as
    SchedulerServiceImpl::WaitForStartingTask makes them (scheduler_service_impl.cc:234-264).
    cpu_baseline leg: the oracle is the thing timed, never the product."""
    from oracle import refbind as R
    from yadcc_amd import synth
    if not R.available():
        return {"error": "oracle/_ref is not built"}
    calls = calls or (2048 if n_servants <= 4000 else 512)
    sv = synth.make_servants(n_servants, n_tasks_hint=70 * n_servants, n_envs=4, seed=42)
    tk = synth.make_tasks(calls, sv, n_envs=4, self_frac=0.0)
    d = R.RefDispatcher()
    d.load_servants(sv)
    d.dispatch_batch({k: v[:64] for k, v in tk.items()})  # warm
    _, _, _, lat = d.dispatch_batch(tk, want_latency=True)
    d.close()
    us = lat.astype(np.float64) / 1e3
    rec = {"kind": "reference", "cores": 1, "calls": int(calls),
           "single_1": {"p50": percentile(us, 0.5), "p99": percentile(us, 0.99), "mean": float(us.mean())}}
    for n in (2, 4, 8, 16, 32, 64, 128, 256):
        if calls // n < 4:
            break
        g = us[:calls // n * n].reshape(-1, n).sum(axis=1)
        rec["batch_%d" % n] = {"p50": percentile(g, 0.5), "p99": percentile(g, 0.99), "mean": float(g.mean())}
    return rec


def reference_parked():
    """cpu_baseline leg of the parked-waiters measurement: the reference class itself, multi-threaded
    build (oracle/_ref/ref_parked_bench: real clock, real condition variable), same workload. A
    bounded sample: at 10 000 waiters one FreeTask costs the reference ~0.3 s."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_parked_bench")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/ref_parked_bench is not built"}
    out = {"kind": "reference", "cores": os.cpu_count()}
    for argv in (["60", "1", "100", "1000"], ["12", "1", "10000"]):
        r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if r.returncode != 0:
            return {"error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
        out.update(json.loads(r.stdout.strip().splitlines()[-1])["waiters"])
    return out


def reference_td_surface(n_servants=2000, n_leases=100_000, heartbeats=150):
    """The verbatim reference class (oracle/_ref) on the host: 2k servants, 10^5 live leases;
    KeepServantAlive + NotifyServantRunningTasks of `heartbeats` servants, each reporting the
    grants it holds, then GetRunningTasks. cpu_baseline leg: the oracle is the thing timed here,
    never the product."""
    from oracle import refbind as R
    from yadcc_amd import synth
    if not R.available():
        return {"error": "oracle/_ref is not built"}
    sv = synth.make_servants(n_servants, n_tasks_hint=2 * n_leases, n_envs=4, seed=42)
    tk = synth.make_tasks(n_leases, sv, n_envs=4)
    d = R.RefDispatcher()
    d.load_servants(sv)
    idx, ids, wait_secs, _ = d.dispatch_batch(tk)
    ok = idx < R.IDX_ENV_NOT_FOUND
    order = np.argsort(idx[ok], kind="stable")
    held_idx, held_ids = idx[ok][order], ids[ok][order]
    starts = np.searchsorted(held_idx, np.arange(n_servants + 1))
    who = np.linspace(0, n_servants - 1, heartbeats).astype(np.int64)
    rows = [{k: v[s:s + 1] for k, v in sv.items()} for s in who]
    locs = ["%d.%d.%d.%d:%d" % (ip >> 24, (ip >> 16) & 255, (ip >> 8) & 255, ip & 255, port)
            for ip, port in zip(sv["ip"][who].tolist(), sv["port"][who].tolist())]
    reported = 0
    t0 = time.perf_counter()
    for s, row, loc in zip(who, rows, locs):
        d.load_servants(row)  # KeepServantAlive of a known location (renewal)
        mine = held_ids[starts[s]:starts[s + 1]]
        unknown = d.notify_servant_running_tasks(loc, mine)
        reported += len(mine) - len(unknown)
    hb_secs = time.perf_counter() - t0
    polls = 10
    t0 = time.perf_counter()
    for _ in range(polls):
        listed = len(d.get_running_tasks(cap=n_leases))
    poll_secs = time.perf_counter() - t0
    # OnExpirationTimer with that many leases in the table (it walks all of them, :523-535)
    t0 = time.perf_counter()
    for _ in range(5):
        R.fire_timers()
    timer_ms = 1e3 * (time.perf_counter() - t0) / 5
    d.close()
    return {"kind": "reference", "cores": 1, "servants": n_servants, "leases": int(ok.sum()),
            "wait_for_starting_new_task_per_s": float(ok.sum()) / wait_secs,
            "heartbeats_per_s": heartbeats / hb_secs, "us_per_heartbeat": 1e6 * hb_secs / heartbeats,
            "reported_tasks_per_heartbeat": reported / heartbeats,
            "get_running_tasks_per_s": polls / poll_secs, "running_tasks_listed": listed,
            "expiration_timer_tick_ms": timer_ms,
            "sample": "%d heartbeats (KeepServantAlive + NotifyServantRunningTasks), %d GetRunningTasks "
                      "calls, %.1f s" % (heartbeats, polls, hb_secs + poll_secs)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # A step is ~0.1 ms: enough of them that one scheduling hiccup of the host does not move
    # the mean, and that p99 means something.
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--shared-ip-frac", type=float, default=0.0,
                    help="fraction of servants that share a host with an earlier one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--digests", type=int, default=0,
                    help="number of distinct compiler digests in the pool instead of the configuration's "
                         "(e.g. 150: every servant advertises its own few, ~one servant class per servant)")
    ap.add_argument("--resident-only", action="store_true",
                    help="skip the host-buffer (end_to_end) loops: what tools/profile.sh runs under rocprofv3, "
                         "so that per-kernel averages are those of the HBM-resident batches")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one synchronous ydc_dispatch_device call per step instead of two batches in flight")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="only the selected configuration: no `configs` records (N = 1), no `strong_cfg4` "
                         "record (N > 1), no steady-state COMMIT loop")
    ap.add_argument("--transport", choices=("auto", "rccl", "ipc", "ipc-host"), default="auto",
                    help="N > 1: how the ranks exchange boundary states and slot deltas. auto = RCCL "
                         "(the default transport), and if its communicator does not come up in time "
                         "the RCCL-free mailbox transport of libydc.so (HIP IPC device memory, then "
                         "the shared host segment) instead of giving the sharded run up")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher of N
    ranks of itself (what torch.distributed.run would do: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in the environment, gloo rendezvous on loopback). Rank 0's stdout
    is captured and its ONE JSON line re-printed last; everything else passes through."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   YDC_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))
    deadline = time.time() + float(os.environ.get("YDC_BENCH_LAUNCH_TIMEOUT", "1500"))
    out0 = None
    rcs = [None] * n
    try:
        # (rank 0's pipe is drained by communicate(); the others write nothing to stdout)
        try:
            out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            out0 = None
        for r, p in enumerate(procs):
            try:
                rcs[r] = p.wait(timeout=max(1.0, deadline - time.time()) if out0 is not None else 1.0)
            except subprocess.TimeoutExpired:
                rcs[r] = None
    finally:
        for p in procs:
            if p.poll() is None:  # exactly the processes started here, by pid / process group
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    p.kill()
    text = (out0 or b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    for ln in text.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(rc != 0 for rc in rcs) or len(lines) != 1:
        print("[bench] self-launched ranks failed: exit codes %s, %d JSON lines" % (rcs, len(lines)),
              file=sys.stderr)
        return 1
    j = json.loads(lines[0])
    if j.get("n_gpus") != n:
        print("[bench] the ranks report n_gpus = %r, asked for %d" % (j.get("n_gpus"), n), file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def setup(args):
    """Rank environment, the gloo group (N > 1), this rank's context and — N > 1 — the group
    transport of libydc.so. Returns the run's state."""
    E = types.SimpleNamespace()
    E.rank = int(os.environ.get("RANK", 0))
    E.world = int(os.environ.get("WORLD_SIZE", 1))
    E.local_rank = int(os.environ.get("LOCAL_RANK", 0))
    E.use_dist = E.world > 1 or os.environ.get("YDC_BENCH_FORCE_DIST") == "1"
    E.dist = E.torch = None
    E.t_start = time.perf_counter()
    E.group_note = None
    E.sharded = False
    E.init_thread = None
    E.rccl_ranks, E.is_rccl = 0, False
    E.transport = "none"
    E.abandoned = []  # contexts stuck inside a communicator bootstrap: never torn down

    def phase(what):
        # Start-up phases of a distributed run on stderr: the rendezvous and the RCCL bootstrap
        # are the parts whose duration depends on the box, not on this code.
        if E.use_dist and E.rank == 0:
            print("[bench] %6.1f s  %s" % (time.perf_counter() - E.t_start, what), file=sys.stderr, flush=True)

    E.phase = phase
    if E.use_dist:
        # Rendezvous, barriers and the max over ranks go through gloo (CPU); the data path of
        # the sharded batch is RCCL inside libydc.so (ydc_group_init / ydc_dispatch_sharded).
        import torch
        import torch.distributed as dist
        E.torch, E.dist = torch, dist
        phase("torch imported")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if E.world == 1:  # (YDC_BENCH_FORCE_DIST=1 without a launcher: a group of one rank)
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        import datetime
        # (a rank that died leaves the others in a collective: give up after 15 minutes, not gloo's 30)
        dist.init_process_group("gloo", rank=E.rank, world_size=E.world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("YDC_BENCH_GLOO_TIMEOUT", "900"))))
        phase("gloo group up (%d ranks)" % E.world)

    from yadcc_amd import binding
    # One rank per GPU; if the launcher narrowed this process to a single visible device it is
    # 0, and with more ranks than devices (the 1-GPU test box) ranks share devices.
    E.n_dev = binding.device_count()
    E.device = E.local_rank % E.n_dev if E.n_dev else E.local_rank
    E.ranks_per_device = -(-E.world // E.n_dev) if E.n_dev else 1
    E.ctx = binding.Context(device=E.device)
    if not E.use_dist:
        return E
    torch, dist = E.torch, E.dist

    def everybody(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0]) == 1

    want_rccl = args.transport in ("auto", "rccl")
    if want_rccl and args.transport == "auto" and E.ranks_per_device > 1:
        # RCCL refuses two ranks on one device: do not wait for its bootstrap to time out.
        want_rccl = False
        E.group_note = "RCCL skipped: %d ranks on %d device(s)" % (E.world, E.n_dev)
    if want_rccl:
        # One node: RCCL's bootstrap only has to find the loopback interface (probing the
        # other interfaces / InfiniBand takes minutes on some boxes); the data path is xGMI.
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        ok = False
        try:
            ids = [binding.group_unique_id() if E.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            # The communicator bootstrap is a blocking C call: give it a deadline instead
            # of hanging the whole scaling run (the thread is abandoned if it never returns).
            import threading
            box = {}
            ctx0 = E.ctx

            def _init():
                try:
                    ctx0.group_init(ids[0], E.rank, E.world)
                    box["ok"] = True
                except Exception as e:  # noqa: BLE001
                    box["err"] = e

            phase("RCCL unique id shared")
            th = E.init_thread = threading.Thread(target=_init, daemon=True)
            t_rccl = time.perf_counter()
            th.start()
            th.join(float(os.environ.get("YDC_BENCH_RCCL_TIMEOUT", "240")))
            ok = bool(box.get("ok"))
            if not ok:
                E.group_note = "RCCL group init failed after %.0f s (%s)" % (
                    time.perf_counter() - t_rccl,
                    box.get("err") or "ncclCommInitRank did not return in time")
        except Exception as e:  # noqa: BLE001  (keep the scaling run alive, say what happened)
            E.group_note = "RCCL group init failed (%s)" % e
        all_ok = everybody(ok)
        if all_ok:
            E.sharded, E.transport = True, "rccl"
            E.rccl_ranks, E.is_rccl = E.ctx.group_size()  # ncclCommCount of the communicator
        else:
            E.group_note = E.group_note or "another rank could not join the RCCL group"
            if E.init_thread is not None and E.init_thread.is_alive():
                # still inside ncclCommInitRank: leave that context alone, take a fresh one
                E.abandoned.append(E.ctx)
                E.ctx = binding.Context(device=E.device)
            elif ok:
                E.ctx.group_destroy()
        phase(E.group_note or "RCCL communicator up")
    if not E.sharded and args.transport != "rccl":
        # The RCCL-free transport: mailboxes written by the peers' kernels (ydc_group_ipc_export
        # / ydc_group_init_ipc), handles all-gathered over gloo. Same protocol, same results.
        # (every rank takes part in every collective below, whatever happened to it locally)
        note = E.group_note
        try:
            mine = E.ctx.group_ipc_export(E.rank, E.world)
        except binding.YdcError as e:
            mine = None
            note = (note + "; " if note else "") + str(e)
        handles = [None] * E.world
        dist.all_gather_object(handles, mine)
        kinds = [binding.TRANSPORT_IPC_DEVICE, binding.TRANSPORT_IPC_HOST]
        if args.transport == "ipc-host":
            kinds = kinds[1:]
        for kind in kinds if all(h is not None for h in handles) else []:
            try:
                E.ctx.group_init_ipc(handles, E.rank, E.world, kind)
                ok = True
            except binding.YdcError as e:
                ok = False
                note = (note + "; " if note else "") + str(e)
            if everybody(ok):
                E.sharded, E.transport = True, binding.TRANSPORT_NAMES[kind]
                break
        E.group_note = note
        phase("mailbox transport: %s" % (E.transport if E.sharded else "unavailable"))
    if not E.sharded:
        E.group_note = (E.group_note or "no transport") + ": ranks ran independent batches"
    return E


def measure_config(E, args, config, scaling, steps, warmup, detail, digests=0, cpu_prefix=0):
    """Times `steps` batches of `config` on the run's context(s). detail "full": the driver's
    line (end-to-end loops, CPU baseline, >= 100 latency samples); "compact": a sub-record of
    it (resident loop, synchronous latencies, per-kernel events, parity against the committed
    fixture). Returns the record on rank 0, None elsewhere."""
    from yadcc_amd import binding, pack, synth
    dist, torch, ctx = E.dist, E.torch, E.ctx
    rank, world = E.rank, E.world
    full = detail == "full"

    def barrier():
        if dist:
            dist.barrier()

    # Weak: one global batch of G x (the config's requests) on G x (the config's servants).
    # Strong: the config's own batch and pool. Either way rank r owns the r-th range of the
    # batch (arrival order).
    n_cfg, s_cfg, n_envs, unk = synth.CONFIGS[config]
    n_envs = (args.digests if full else digests) or n_envs
    shared = args.shared_ip_frac if full else 0.0
    mult = world if scaling == "weak" else 1
    n_all = n_cfg * mult
    sv = synth.make_servants(s_cfg * mult, n_tasks_hint=n_all, n_envs=n_envs, seed=42,
                             shared_ip_frac=shared)
    tk_all = synth.make_tasks(n_all, sv, n_envs=n_envs, unknown_env_frac=unk)
    lo, hi = n_all * rank // world, n_all * (rank + 1) // world
    tk = {k: v[lo:hi] for k, v in tk_all.items()}
    n_tasks, n_serv = hi - lo, len(sv["version"])
    ctx.upload_servants(pack.to_abi_columns(sv))
    DA = binding.DeviceArray
    dev = E.device
    d_env = DA.from_numpy(tk["env_id"], dev)
    d_minv = DA.from_numpy(tk["min_version"], dev)
    d_ip = DA.from_numpy(tk["requestor_ip"], dev)
    d_out = DA(n_tasks, np.uint32, dev)
    d_run = DA(n_serv, np.uint32, dev)
    sharded = E.sharded

    def step(commit=False):
        # Returns after the batch's results are final in HBM (stream sync inside).
        if sharded:
            ctx.dispatch_sharded(d_env, d_minv, d_ip, d_out, None, d_run, commit=commit)
        else:
            ctx.dispatch_device(d_env, d_minv, d_ip, d_out, None, d_run, commit=commit)

    # One GPU: the steps are pipelined two deep (ydc_dispatch_device_async / ydc_dispatch_wait):
    # batch k + 1 is enqueued before the host looks at the outcome of batch k, each into its own
    # result buffers. Every step still places the whole batch and waits for its results inside
    # the timed region; what disappears is the device idling while the host turns around.
    # (--no-pipeline: one synchronous call per step, as in rounds 1 and 2.)
    pipelined = not sharded and not E.use_dist and not args.no_pipeline and steps >= 2
    d_out2 = DA(n_tasks, np.uint32, dev) if pipelined else None
    d_run2 = DA(n_serv, np.uint32, dev) if pipelined else None

    def run_steps(k):
        if k <= 0:
            return
        if not pipelined:
            for _ in range(k):
                step()
            return
        ctx.dispatch_device_async(d_env, d_minv, d_ip, d_out, None, d_run)
        for i in range(1, k):
            if i & 1:
                ctx.dispatch_device_async(d_env, d_minv, d_ip, d_out2, None, d_run2)
            else:
                ctx.dispatch_device_async(d_env, d_minv, d_ip, d_out, None, d_run)
            ctx.dispatch_wait()
        ctx.dispatch_wait()

    run_steps(warmup)
    ctx.synchronize()
    barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-batch latency (p50 / p99) and the synchronous rate: one batch in flight, a few untimed
    # calls first (the switch from the pipelined loop, a short run's clocks still settling), at
    # least n_lat samples whatever --steps is.
    n_lat = 100  # per-batch wall time over >= 100 repeats, sub-records included
    lat = []
    if pipelined:
        for _ in range(5):
            step()
        for _ in range(max(n_lat, min(steps, 1000)) if full else n_lat):
            s0 = time.perf_counter()
            step()
            lat.append(time.perf_counter() - s0)
        step()  # (d_out / d_run hold a synchronous batch's results for the checks below)
    sync_ms = 1e3 * sum(lat) / len(lat) if lat else None
    if not pipelined:
        # (N > 1: the timed steps are synchronous; their own latencies serve)
        barrier()
        for _ in range(min(steps, n_lat)):
            s0 = time.perf_counter()
            step()
            lat.append(time.perf_counter() - s0)
    st = ctx.stats()
    granted_all = float(st["granted"])
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = torch.tensor([granted_all], dtype=torch.float64)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        elapsed, granted_all = float(t[0]), float(g[0])
    # p50 / p99 want >= 100 samples on the driver's line whatever --steps is (extra batches,
    # outside the timed region).
    lat_all = list(lat)
    while len(lat_all) < n_lat:
        barrier()
        s0 = time.perf_counter()
        step()
        lat_all.append(time.perf_counter() - s0)

    # SURVEY.md §8(d)'s assignments/s: batch visible to the dispatcher -> all results visible to
    # the host, through the host-buffer entry point ydc_dispatch (H2D of 12 B/request over
    # PCIe, kernels, D2H of 4 B/request). Reported in `end_to_end`, next to `value`.
    e2e = None
    if not E.use_dist and not args.resident_only:
        # (columns and the result array are the caller's and stay the same from batch to batch,
        # as in a scheduler loop; allocating them per call would time numpy, not the dispatch)
        # The scheduler's buffers are page-locked once (ydc_host_alloc): ydc_dispatch then reads
        # the columns and writes the placement in place — no staging copy on either side. The
        # same call with pageable numpy arrays (staged through the library's own pinned arenas)
        # is reported beside it (driver's line only).
        def host_loop(cols, res):
            ctx.dispatch(cols, want_util=False, want_running=False, out_idx=res)
            hl = []
            for _ in range(max(100, min(1000, steps)) if full else n_lat):
                s0 = time.perf_counter()
                ctx.dispatch(cols, want_util=False, want_running=False, out_idx=res)
                hl.append(time.perf_counter() - s0)
            return hl, {"assignments_per_s": st["granted"] * len(hl) / sum(hl),
                        "ms_per_batch": 1e3 * sum(hl) / len(hl),
                        "p50_ms": 1e3 * percentile(hl, 0.50), "p99_ms": 1e3 * percentile(hl, 0.99),
                        "batches": len(hl)}

        tk_c = {k: np.ascontiguousarray(v, dtype=np.uint32) for k, v in tk.items()}
        tk_p = {k: binding.pinned_empty(len(v), np.uint32) for k, v in tk_c.items()}
        for k in tk_c:
            tk_p[k][:] = tk_c[k]
        res_p = binding.pinned_empty(n_tasks, np.uint32)
        if full:
            res = np.empty(n_tasks, np.uint32)
            _, pageable = host_loop(tk_c, res)
        _, e2e = host_loop(tk_p, res_p)
        e2e["definition"] = ("ydc_dispatch with the caller's page-locked host buffers (ydc_host_alloc), "
                             "batch visible to the dispatcher -> placement visible to the host: the "
                             "kernels read the columns and write the results over PCIe in place, "
                             "SURVEY.md 8(d)")
        if full:
            e2e["same_placement_as_pageable"] = bool(np.array_equal(res_p, res))
            pageable["definition"] = ("the same call with pageable numpy arrays: staged through the "
                                      "library's pinned arenas (two host memcpys + two copy commands)")
            e2e["pageable_buffers"] = pageable
        else:
            e2e["same_placement_as_resident"] = bool(np.array_equal(res_p, d_out.numpy()))

    # Steady state (driver's line, one GPU): every batch COMMITs — the next batch sees the
    # previous one's grants, the dependency a scheduler has between batches — and the grants are
    # released again (ydc_release_slots, what FreeTask feeds the device) before the next one.
    steady = None
    if full and not E.use_dist and not args.no_extra_configs:
        granted_idx = d_out.numpy()
        granted_idx = granted_idx[granted_idx < binding.IDX_ENV_NOT_FOUND]
        run0 = ctx.get_running()
        k_st = max(20, min(200, steps))
        for _ in range(3):
            step(commit=True)
            ctx.release_slots(granted_idx)
        ctx.synchronize()
        s0 = time.perf_counter()
        for _ in range(k_st):
            step(commit=True)
            ctx.release_slots(granted_idx)
        ctx.synchronize()
        dt = time.perf_counter() - s0
        # ... and with the grants given back from where the batch left them: the placement array in
        # HBM (ydc_release_slots_device; entries that are no servant are skipped).
        for _ in range(3):
            step(commit=True)
            ctx.release_slots_device(d_out)
        ctx.synchronize()
        s1 = time.perf_counter()
        for _ in range(k_st):
            step(commit=True)
            ctx.release_slots_device(d_out)
        ctx.synchronize()
        dt_dev = time.perf_counter() - s1
        steady = {"ms_per_step": 1e3 * dt / k_st, "steps": k_st,
                  "ms_per_step_released_from_hbm": 1e3 * dt_dev / k_st,
                  "assignments_per_s": len(granted_idx) * k_st / dt,
                  "registry_restored": bool(np.array_equal(ctx.get_running(), run0)),
                  "same_placement": bool(np.array_equal(
                      d_out.numpy()[d_out.numpy() < binding.IDX_ENV_NOT_FOUND], granted_idx)),
                  "definition": "synchronous ydc_dispatch_device with YDC_DISPATCH_COMMIT followed by "
                                "ydc_release_slots of the batch's grants (%d servant indexes from the "
                                "host), every step" % len(granted_idx)}

    # N > 1: the placement of the whole batch against the oracle (one more step, gathered).
    parity_oracle = None
    if E.use_dist and dist:
        step()
        parts = [None] * world if rank == 0 else None
        dist.gather_object(d_out.numpy(), parts, dst=0)
        if rank == 0:
            from oracle import oraclebind as O
            want, _, wrun = O.dispatch(sv, tk_all, "sorted", want_util=False)
            parity_oracle = bool(np.array_equal(np.concatenate(parts), want) and
                                 np.array_equal(d_run.numpy(), wrun))

    # Per-kernel durations: HIP events on the dispatch stream, separate profiled steps so
    # the events do not perturb the timed region.
    ctx.set_profiling(True)
    per_step = {}
    n_prof = max(3, min(40, steps)) if full else 3
    for _ in range(n_prof):
        step()
        for k, (cnt, ms) in ctx.kernel_profile().items():
            per_step.setdefault(k, []).append((cnt, ms))
    stage_ms = ctx.stats()["stage_ms"]
    ctx.set_profiling(False)
    # Median over the profiled steps (a stalled step must not pass for a slow kernel), scaled
    # back to totals so that the arithmetic below stays "total / count".
    prof = {}
    for k, v in per_step.items():
        cnt = int(np.median([c for c, _ in v]))
        prof[k] = [cnt * n_prof, float(np.median([m for _, m in v])) * n_prof]

    if rank != 0:
        return None
    host_idx = d_out.numpy()
    host_run = d_run.numpy()
    shape = "%d pending requests x %d servants" % (n_all, n_serv)
    out = {
        "metric": METRIC,
        "value": granted_all * steps / elapsed,
        "unit": "assignments/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u32" if st["key_bits"] <= 32 else "u64",
        "data": "synthetic",
        "value_definition": "HBM-resident: request columns, servant table and results stay in "
                            "HBM (kernels + one 200-byte outcome read-back per batch)%s; the "
                            "host-buffer rate of SURVEY.md 8(d) is in end_to_end" % (
                                ", two batches in flight (ydc_dispatch_device_async: the next "
                                "batch is enqueued before the host reads the previous outcome; "
                                "ms_per_step_synchronous = one batch at a time)" if pipelined else ""),
        "pipeline_depth": 2 if pipelined else 1,
        "ms_per_step_synchronous": sync_ms,
        "config": {"workload": "%s%s%s: %s%s, %d classes" % (
                       config, " with %d digests" % n_envs if (full and args.digests) or digests else "",
                       "" if world == 1 else " (%s scaling)" % scaling, shape,
                       ", %.0f %% of the servants on shared hosts" % (100 * shared) if shared else "",
                       st["n_classes"]),
                   "parallelism": "1 GPU" if world == 1 else
                                  ("one global batch of %d requests x %d servants sharded by "
                                   "rank range over %d ranks on %d GPU(s), %s all-gather of boundary "
                                   "states and servant-slot deltas" % (
                                       n_all, n_serv, world, min(world, max(E.n_dev, 1)),
                                       "RCCL" if E.transport == "rccl" else "mailbox (%s)" % E.transport)
                                   if sharded else E.group_note),
                   "inputs": "request columns + servant table resident in HBM; results in HBM"},
        "p99_dispatch_latency_ms": 1e3 * percentile(lat_all, 0.99),
        "p50_dispatch_latency_ms": 1e3 * percentile(lat_all, 0.50),
        "latency_samples": len(lat_all),
        "stats": {k: v for k, v in st.items() if k != "stage_ms"},
        "granted_all_ranks": int(granted_all),
        "kernels_us_per_step": {k: 1e3 * v[1] / n_prof for k, v in prof.items()},
        "kernel_launches_per_step": {k: v[0] / n_prof for k, v in prof.items()},
    }
    # The three rates side by side, whichever of them `value` is.
    out["value_pipelined"] = out["value"] if pipelined else None
    out["value_synchronous"] = (st["granted"] / (sync_ms * 1e-3) if sync_ms else
                                (None if pipelined else out["value"]))
    out["value_end_to_end"] = e2e["assignments_per_s"] if e2e else None
    if full:
        out["stage_ms"] = stage_ms
    if e2e:
        out["end_to_end"] = e2e
        out["host_buffers_assignments_per_s"] = e2e["assignments_per_s"]
    if steady:
        out["steady_state_commit"] = steady
    if E.use_dist:
        # sharded: the N ranks placed ONE global batch through ydc_dispatch_sharded (false:
        # no transport came up and every rank placed its own batch — not a scaling result).
        out["sharded"] = bool(sharded)
        out["transport"] = {"ipc-host": "ipc"}.get(E.transport, E.transport)
        out["transport_detail"] = E.transport + ("" if not E.group_note else " (%s)" % E.group_note)
        out["rccl_ranks"] = E.rccl_ranks
        out["rccl"] = E.is_rccl
        out["parity_vs_oracle"] = parity_oracle
        out["devices"] = E.n_dev
        out["ranks_per_device"] = E.ranks_per_device
    # Pinned to the verbatim reference without a CPU replay: the committed placement of the
    # first 50k requests of this very batch (a sequential batch's prefix is the prefix batch),
    # and conservation: running_tasks after the batch - before = histogram of the placement.
    if world == 1:
        ref = fixture_prefix(config, sv, tk_all)
        if ref is not None:
            out["parity_vs_reference_fixture"] = bool(np.array_equal(host_idx[:len(ref)], ref))
            out["fixture_requests"] = int(len(ref))
            ok, n_dig = fixture_digests_ok(config, sv, tk_all, host_idx)
            if ok is not None:  # (block digests of a longer prefix)
                out["parity_vs_reference_fixture"] = out["parity_vs_reference_fixture"] and ok
                out["fixture_requests"] = max(int(len(ref)), n_dig)
        ok = host_idx < binding.IDX_ENV_NOT_FOUND
        out["conservation"] = bool(np.array_equal(
            host_run.astype(np.int64) - sv["running_tasks"].astype(np.int64),
            np.bincount(host_idx[ok], minlength=n_serv)))
    if digests and world == 1:
        from oracle import oraclebind as O
        want, _, wrun = O.dispatch(sv, tk_all, "sorted", want_util=False)
        out["parity_vs_oracle"] = bool(np.array_equal(host_idx, want) and np.array_equal(host_run, wrun))
    if prof:
        dom = max(prof, key=lambda k: prof[k][1])
        launches, total_ms = prof[dom]
        # bytes(batch) = 16 N + 40 S; one launch of the dominant kernel
        # (a matching pass) works on the whole batch of this rank.
        alg_bytes = 16 * n_tasks + 40 * n_serv
        avg_launch_s = (total_ms / launches) * 1e-3
        ach = alg_bytes / avg_launch_s / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": 8000.0,
                           "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": pmc_traffic(dom, config),
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "avg_launch_us": avg_launch_s * 1e6,
                           "launches_per_step": launches / n_prof,
                           # The same bytes against the whole step (all kernels + host
                           # turn-around): what a batch achieves, whatever the launch count.
                           "per_step_GBps": alg_bytes / (elapsed / steps) / 1e9}
        if full:
            out["roofline"]["note"] = ("the dominant kernel is bound by the dependency chain of the "
                                       "greedy merge, not by bandwidth: see DESIGN.md 3.4")
    if world == 1 and not args.no_cpu_baseline and (full or cpu_prefix):
        # cfg2: the whole batch. cfg3 / cfg4: the first 50k requests, the rate taken
        # as the batch's (the reference only gets slower as servants fill up: full runs take
        # ~2 min / ~20 min).
        ref_idx, base = cpu_baseline(sv, tk, max_tasks=100_000 if full else cpu_prefix)
        base["requests"] = int(len(ref_idx))
        out["cpu_baseline"] = base
        out["parity_vs_cpu_baseline"] = bool(np.array_equal(ref_idx, host_idx[:len(ref_idx)]))
    return out


COMPACT_KEYS = ("value", "unit", "ms_per_step", "ms_per_step_synchronous", "steps", "warmup", "scaling",
                "n_gpus", "dtype", "p50_dispatch_latency_ms", "p99_dispatch_latency_ms",
                "latency_samples", "kernels_us_per_step", "kernel_launches_per_step",
                "parity_vs_reference_fixture", "fixture_requests", "fixture_ticks", "conservation",
                "parity_vs_oracle", "sharded", "transport", "value_end_to_end", "cpu_baseline",
                "parity_vs_cpu_baseline")


def compact(rec):
    """A sub-record of the driver's line: what was timed, how long it took, whether it is right."""
    if rec is None:
        return None
    c = {k: rec[k] for k in COMPACT_KEYS if k in rec}
    c["workload"] = rec["config"]["workload"]
    c["rounds"] = rec["stats"].get("rounds")
    c["granted"] = rec["stats"].get("granted")
    if "tick_assembled_in_place" in rec:
        c["tick_assembled_in_place"] = {k: rec["tick_assembled_in_place"][k] for k in ("ms_per_step", "p99_ms", "ticks")}
    if rec.get("tick_enqueued_eagerly"):
        c["tick_enqueued_eagerly"] = {k: rec["tick_enqueued_eagerly"][k] for k in ("ms_per_step", "p99_ms", "ticks",
                                                                                  "parity_vs_reference_fixture")}
    if "end_to_end" in rec:
        c["end_to_end_ms"] = rec["end_to_end"]["ms_per_batch"]
        c["end_to_end_p99_ms"] = rec["end_to_end"]["p99_ms"]
    if "roofline" in rec:
        c["roofline"] = {k: rec["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic",
                                                         "avg_launch_us", "launches_per_step",
                                                         "algorithmic_bytes_per_launch")
                         if k in rec["roofline"]}
    return c


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 8192  # the driver keeps a bounded tail of stdout: a line it cannot ingest is no measurement


def r4(x):
    """Four significant digits for the sub-records of the line (the contract keys stay exact)."""
    if isinstance(x, float):
        return float("%.4g" % x)
    if isinstance(x, dict):
        return {k: r4(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [r4(v) for v in x]
    return x


def driver_line(full):
    """The ONE line of stdout: the bench contract's keys, `roofline`, `cpu_baseline`, and one short
    record per BASELINE.json configuration. Everything else that was measured in this run (per-kernel
    times, the latency tables of the TaskDispatcher surface, concurrent callers, the pageable-buffer
    variants, ...) is in DETAIL_FILE, written next to this script; the line names it."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")
    line = {k: full[k] for k in keep}
    if "roofline" in full:
        line["roofline"] = {k: full["roofline"][k] for k in (
            "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us",
            "algorithmic_bytes_per_launch") if k in full["roofline"]}
    if "cpu_baseline" in full:
        line["cpu_baseline"] = {k: full["cpu_baseline"][k] for k in (
            "value", "unit", "cores", "kind", "sample", "p99_latency_us", "soa_port_value")
            if k in full["cpu_baseline"]}
    side = {}
    for k in ("value_definition", "value_synchronous", "value_end_to_end", "ms_per_step_synchronous",
              "p50_dispatch_latency_ms", "p99_dispatch_latency_ms", "latency_samples", "granted_all_ranks",
              "parity_vs_cpu_baseline", "parity_vs_reference_fixture", "fixture_requests", "fixture_ticks",
              "parity_ticks", "conservation", "parity_vs_oracle", "sharded", "transport", "rccl_ranks"):
        if k in full:
            side[k] = full[k]
    if "end_to_end" in full:
        side["end_to_end_ms"] = full["end_to_end"]["ms_per_batch"]
        side["end_to_end_p99_ms"] = full["end_to_end"]["p99_ms"]
    if "steady_state_commit" in full:
        side["commit_and_release_ms_per_step"] = full["steady_state_commit"]["ms_per_step"]
    line.update(r4(side))

    def short(c):
        s = {k: c[k] for k in ("ms_per_step", "value", "value_end_to_end", "p50_dispatch_latency_ms",
                               "p99_dispatch_latency_ms", "latency_samples", "steps",
                               "parity_vs_reference_fixture", "fixture_requests", "fixture_ticks",
                               "parity_vs_oracle", "conservation", "n_gpus", "scaling", "sharded")
             if c.get(k) is not None}
        if "roofline" in c:
            s["roofline"] = {k: c["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_us", "traffic")}
        if "cpu_baseline" in c:
            s["cpu_baseline"] = {k: c["cpu_baseline"][k] for k in ("value", "kind", "cores", "requests")
                                 if k in c["cpu_baseline"]}
            s["parity_vs_cpu_baseline"] = c.get("parity_vs_cpu_baseline")
        return r4(s)

    if "configs" in full:
        line["configs"] = {k: short(v) for k, v in full["configs"].items() if v}
    if full.get("strong_cfg4"):
        line["strong_cfg4"] = short(full["strong_cfg4"])
    td = full.get("td_surface") or {}
    if td and "error" not in td:
        s = {}
        for S in (2000, 16000):
            lat = td.get("latency", {}).get("servants_%d" % S, {})
            one = lat.get("tick", {}).get("single_1")
            ref = lat.get("reference", {}).get("single_1") if isinstance(lat.get("reference"), dict) else None
            if isinstance(one, dict):
                s["call_us_%dk_servants" % (S // 1000)] = {
                    "p50": one.get("p50"), "p99": one.get("p99"), "p999": one.get("p999"),
                    "reference_p50": ref and ref.get("p50"), "reference_p99": ref and ref.get("p99")}
        for name, short in (("timer_2k_servants_100k_leases", "timer_on_100k_leases_us"),
                            ("timer_16k_servants_1M_leases", "timer_on_1M_leases_us")):
            t = td.get(name, {}).get("timer_on")
            if isinstance(t, dict):
                s[short] = {k: t.get(k) for k in ("p50", "p99", "p999", "max", "timer_max_us")}
        pw, pr = td.get("parked_waiters", {}), td.get("parked_waiters_reference", {})
        if isinstance(pw, dict) and "error" not in pw:
            s["parked_waiters"] = {
                k: {"wake_to_grant_us_p50": v["wake_to_grant_us"]["p50"], "wake_to_grant_us_p99": v["wake_to_grant_us"]["p99"],
                    "frees_per_s": v["frees_per_s"],
                    "reference_wake_us_p50": pr.get(k, {}).get("wake_to_grant_us", {}).get("p50"),
                    "reference_frees_per_s": pr.get(k, {}).get("frees_per_s")}
                for k, v in pw.items() if isinstance(v, dict)}
        cc = td.get("concurrent_callers_2k_servants")
        if isinstance(cc, dict) and "error" not in cc:
            s["concurrent_calls_per_s"] = {k: v.get("calls_per_s") for k, v in cc.items() if isinstance(v, dict)}
        line["td_surface"] = r4(s)
    line["detail_file"] = DETAIL_FILE
    return line


def emit(full):
    """Writes everything measured to DETAIL_FILE and prints the driver's line (< LINE_LIMIT bytes,
    asserted: a line the driver cannot parse is an unmeasured round — BENCH_r05)."""
    line = driver_line(full)
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:  # (a read-only checkout: the line still goes out)
        line["detail_file"] = "not written: %s" % e
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:  # never reached with today's records; degrade rather than lose the round
        for k in ("td_surface", "strong_cfg4", "configs"):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    print(text, flush=True)


def main():
    args = parse_args()
    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        # Never print a line whose n_gpus is not what was asked for.
        print("[bench] --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to run"
              % (args.gpus, world), file=sys.stderr)
        return 2
    if args.config == "cfg5":
        if world != 1:
            print("[bench] cfg5 (streaming) is a single-GPU configuration", file=sys.stderr)
            return 2
        out = stream_record(args, args.steps, args.warmup, 0 if args.no_cpu_baseline else 30,
                            device=int(os.environ.get("LOCAL_RANK", 0)))
        emit(out)
        return 0

    E = setup(args)
    out = measure_config(E, args, args.config, args.scaling, args.steps, args.warmup, "full")
    extra = not args.no_extra_configs and not args.digests and not args.shared_ip_frac
    if extra and E.world > 1 and E.sharded and not (args.config == "cfg4" and args.scaling == "strong"):
        # BASELINE.json configs[3] as specified: 4M requests x 16k servants cut over the ranks.
        E.phase("%s (%s) timed; cfg4 strong next" % (args.config, args.scaling))
        k = max(3, min(args.steps, 20))
        rec = measure_config(E, args, "cfg4", "strong", k, min(args.warmup, 3), "compact")
        if out is not None:
            out["strong_cfg4"] = compact(rec)
    if extra and E.world == 1 and not E.use_dist and args.config == "cfg2" and not args.resident_only:
        # The other single-GPU configurations of BASELINE.json on the same line.
        out["configs"] = {}
        for cfg in ("cfg3", "cfg4"):
            k = max(5, min(args.steps, 20))
            # (ten untimed batches: the library decides from a shape's first seven whether the walk of
            # the dedicated tier's end pays on this registry — DESIGN.md 9.3)
            out["configs"][cfg] = compact(measure_config(E, args, cfg, "weak", k, 10, "compact",
                                                         cpu_prefix=50_000))
        # Sparse eligibility: cfg2's batch on a pool with 150 digests, every servant advertising its
        # own handful (~one servant class per servant; the reference has no limit on them,
        # task_dispatcher.h:93-94) — the walk in groups of 64 requests (wide_kernel.h).
        rec = measure_config(E, args, "cfg2", "weak", 3, 2, "compact", digests=150)
        out["configs"]["cfg2_150_digests"] = compact(rec)
    if extra and E.world == 1 and not E.use_dist and not args.resident_only:
        E.ctx.close()  # (the native tool opens its own context on the device)
        out["td_surface"] = td_surface(with_reference=not args.no_cpu_baseline)
    line_obj = out
    stuck = E.init_thread is not None and E.init_thread.is_alive()  # still inside ncclCommInitRank
    if E.sharded:
        if E.dist:
            E.dist.barrier()  # nobody unmaps a mailbox a peer may still be writing to
        E.ctx.group_destroy()
    E.ctx.close()
    if line_obj is not None and "configs" in line_obj:
        # (>= 1000 ticks for the p99, SURVEY.md 8d: 0.1 s of device time; the reference replays the
        # first 30 of them beside it: ~4 s)
        rec = stream_record(args, 1000, 50, 0 if args.no_cpu_baseline else 30, device=E.device)
        c = compact(rec)
        c["value_definition"] = rec["value_definition"]
        line_obj["configs"]["cfg5"] = c
    # ONE JSON line, and the last thing on stdout: RCCL prints a version banner through C
    # stdio, which would otherwise be flushed behind it at exit. Every rank flushes before the
    # last barrier; rank 0 prints after it.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if E.dist:
        E.dist.barrier()
        E.dist.destroy_process_group()
    if line_obj is not None:
        emit(line_obj)
    if stuck:
        os._exit(0)  # (do not tear the context down under a bootstrap that never returned)
    return 0


def pmc_traffic(kernel, config):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this config
    (profiles/*_<config>_pmc_hbm.json, written by tools/profile.sh: FETCH_SIZE + WRITE_SIZE with
    the calibration factors of profiles/*_hbm_calibration.txt; separate --pmc passes), or None
    when no profile of this kernel is committed. PMC counters cannot be read from inside the
    timed run, so this is the figure of the profiled run of the same command."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_hbm.json" % config))):
        try:
            j = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if kernel in j.get("kernels", {}):
            best = j["kernels"][kernel].get("hbm_bytes_per_launch")
    return best


if __name__ == "__main__":
    sys.exit(main() or 0)
