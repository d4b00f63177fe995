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
which is which.

GPU work goes through yadcc_amd/libydc.so only (its own HIP runtime, /opt/rocm). For
N > 1 (launched by torch.distributed.run, one rank per GPU) torch.distributed (gloo) is used
for the rendezvous, the barriers and the max-over-ranks of the wall time; the data path of
the sharded batch is RCCL inside libydc.so. `--scaling weak` (default): G ranks place ONE
global batch of G x (the config's requests) on a pool of G x (the config's servants);
`--scaling strong`: the config's batch itself (e.g. --config cfg4: 4M requests x 16k
servants = BASELINE.json configs[3]) is cut into G rank ranges.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

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


def stream_main(args):
    """BASELINE.json configs[4]: 10k requests/tick x 2k servants with rolling heartbeats (10 % of
    the servants per tick) and 10k frees per tick; the whole tick is one replay of a captured
    hipGraph (ydc_stream_tick). A step is a tick; only the tick call is timed (the event
    generator is host-side test scaffolding). Host buffers in, host results out. The reference
    class replays the first ticks of the very same stream beside it (heartbeats, frees by
    grant id, sequential WaitForStartingNewTask calls): cpu_baseline + per-tick parity."""
    from yadcc_amd import binding, pack, streaming, synth
    sv, _ = synth.make_config("cfg5")
    es = streaming.EventStream(sv, 10_000, 10_000)
    ctx = binding.Context(device=int(os.environ.get("LOCAL_RANK", 0)))
    ctx.upload_servants(pack.to_abi_columns(sv))
    ctx.stream_begin(es.hb + 8, 10_000, 10_000)
    ref = ref_ids = None
    ref_ticks = 0 if args.no_cpu_baseline else 30  # ~ 300k reference calls, ~10 s
    ref_secs, ref_granted, ref_lat, parity = 0.0, 0, [], True
    if ref_ticks:
        from oracle import refbind as R
        if R.available():
            ref = R.RefDispatcher()
            ref.load_servants(sv)
            ref_ids = np.empty(0, np.uint64)  # grant id of every live grant, stream order
    lat, granted = [], 0
    for t in range(args.warmup + args.steps):
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
        es.commit(got)
        if t >= args.warmup:
            lat.append(dt)
            granted += int((got < binding.IDX_ENV_NOT_FOUND).sum())
    st = ctx.stats()
    out = {
        "metric": METRIC,
        "value": granted / sum(lat), "unit": "assignments/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * sum(lat) / len(lat),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "value_definition": "end to end per tick: host buffers in (one H2D copy), registry "
                            "deltas + batch + commit, results back on the host (PCIe included)",
        "config": {"workload": "cfg5 streaming: 10000 requests + 10000 frees + %d heartbeats per "
                               "tick x %d servants, hipGraph-captured step" % (es.hb, es.n),
                   "parallelism": "1 GPU", "inputs": "host buffers per tick (PCIe included)"},
        "p99_dispatch_latency_ms": 1e3 * percentile(lat, 0.99),
        "p50_dispatch_latency_ms": 1e3 * percentile(lat, 0.50),
        "stats": {k: v for k, v in st.items() if k != "stage_ms"},
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
    print(json.dumps(out))
    ctx.stream_end()
    ctx.close()


def main():
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
    ap.add_argument("--transport", choices=("auto", "rccl", "ipc", "ipc-host"), default="auto",
                    help="N > 1: how the ranks exchange boundary states and slot deltas. auto = RCCL "
                         "(the default transport), and if its communicator does not come up in time "
                         "the RCCL-free mailbox transport of libydc.so (HIP IPC device memory, then "
                         "the shared host segment) instead of giving the sharded run up")
    args = ap.parse_args()
    if args.config == "cfg5":
        return stream_main(args)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    use_dist = world > 1 or os.environ.get("YDC_BENCH_FORCE_DIST") == "1"
    dist = torch = None
    t_start = time.perf_counter()

    def phase(what):
        # Start-up phases of a distributed run on stderr: the rendezvous and the RCCL bootstrap
        # are the parts whose duration depends on the box, not on this code.
        if use_dist and rank == 0:
            print("[bench] %6.1f s  %s" % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)

    if use_dist:
        # Rendezvous, barriers and the max over ranks go through gloo (CPU); the data path of
        # the sharded batch is RCCL inside libydc.so (ydc_group_init / ydc_dispatch_sharded).
        import torch
        import torch.distributed as dist
        phase("torch imported")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if world == 1:  # (YDC_BENCH_FORCE_DIST=1 without a launcher: a group of one rank)
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        phase("gloo group up (%d ranks)" % world)

    from yadcc_amd import binding, pack, synth

    def barrier():
        if dist:
            dist.barrier()

    # Weak: one global batch of G x (the config's requests) on G x (the config's servants).
    # Strong: the config's own batch and pool. Either way rank r owns the r-th range of the
    # batch (arrival order).
    n_cfg, s_cfg, n_envs, unk = synth.CONFIGS[args.config]
    n_envs = args.digests or n_envs
    mult = world if args.scaling == "weak" else 1
    n_all = n_cfg * mult
    sv = synth.make_servants(s_cfg * mult, n_tasks_hint=n_all, n_envs=n_envs, seed=42,
                             shared_ip_frac=args.shared_ip_frac)
    tk_all = synth.make_tasks(n_all, sv, n_envs=n_envs, unknown_env_frac=unk)
    lo, hi = n_all * rank // world, n_all * (rank + 1) // world
    tk = {k: v[lo:hi] for k, v in tk_all.items()}
    n_tasks, n_serv = hi - lo, len(sv["version"])
    # One rank per GPU; if the launcher narrowed this process to a single visible device it is 0.
    n_dev = binding.device_count()
    if n_dev and local_rank >= n_dev:
        local_rank = local_rank % n_dev
    ctx = binding.Context(device=local_rank)
    ctx.upload_servants(pack.to_abi_columns(sv))
    group_note = None
    sharded = False
    init_thread = None
    rccl_ranks, is_rccl = 0, False
    transport = "none"
    abandoned = []  # contexts stuck inside a communicator bootstrap: never torn down
    if use_dist:
        def everybody(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t[0]) == 1

        if args.transport in ("auto", "rccl"):
            # One node: RCCL's bootstrap only has to find the loopback interface (probing the
            # other interfaces / InfiniBand takes minutes on some boxes); the data path is xGMI.
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
            ok = False
            try:
                ids = [binding.group_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                # The communicator bootstrap is a blocking C call: give it a deadline instead
                # of hanging the whole scaling run (the thread is abandoned if it never returns).
                import threading
                box = {}

                def _init():
                    try:
                        ctx.group_init(ids[0], rank, world)
                        box["ok"] = True
                    except Exception as e:  # noqa: BLE001
                        box["err"] = e

                phase("registry uploaded, RCCL unique id shared")
                th = init_thread = threading.Thread(target=_init, daemon=True)
                t_rccl = time.perf_counter()
                th.start()
                th.join(float(os.environ.get("YDC_BENCH_RCCL_TIMEOUT", "240")))
                ok = bool(box.get("ok"))
                if not ok:
                    group_note = "RCCL group init failed after %.0f s (%s)" % (
                        time.perf_counter() - t_rccl,
                        box.get("err") or "ncclCommInitRank did not return in time")
            except Exception as e:  # noqa: BLE001  (keep the scaling run alive, say what happened)
                group_note = "RCCL group init failed (%s)" % e
            all_ok = everybody(ok)
            if all_ok:
                sharded, transport = True, "rccl"
                rccl_ranks, is_rccl = ctx.group_size()  # ncclCommCount of the communicator
            else:
                group_note = group_note or "another rank could not join the RCCL group"
                if init_thread is not None and init_thread.is_alive():
                    # still inside ncclCommInitRank: leave that context alone, take a fresh one
                    abandoned.append(ctx)
                    ctx = binding.Context(device=local_rank)
                    ctx.upload_servants(pack.to_abi_columns(sv))
                elif ok:
                    ctx.group_destroy()
            phase(group_note or "RCCL communicator up")
        if not sharded and args.transport != "rccl":
            # The RCCL-free transport: mailboxes written by the peers' kernels (ydc_group_ipc_export
            # / ydc_group_init_ipc), handles all-gathered over gloo. Same protocol, same results.
            # (every rank takes part in every collective below, whatever happened to it locally)
            try:
                mine = ctx.group_ipc_export(rank, world)
            except binding.YdcError as e:
                mine = None
                group_note = (group_note + "; " if group_note else "") + str(e)
            handles = [None] * world
            dist.all_gather_object(handles, mine)
            kinds = [binding.TRANSPORT_IPC_DEVICE, binding.TRANSPORT_IPC_HOST]
            if args.transport == "ipc-host":
                kinds = kinds[1:]
            for kind in kinds if all(h is not None for h in handles) else []:
                try:
                    ctx.group_init_ipc(handles, rank, world, kind)
                    ok = True
                except binding.YdcError as e:
                    ok = False
                    group_note = (group_note + "; " if group_note else "") + str(e)
                if everybody(ok):
                    sharded, transport = True, binding.TRANSPORT_NAMES[kind]
                    break
            phase("mailbox transport: %s" % (transport if sharded else "unavailable"))
        if not sharded:
            group_note = (group_note or "no transport") + ": ranks ran independent batches"
    DA = binding.DeviceArray
    d_env = DA.from_numpy(tk["env_id"], local_rank)
    d_minv = DA.from_numpy(tk["min_version"], local_rank)
    d_ip = DA.from_numpy(tk["requestor_ip"], local_rank)
    d_out = DA(n_tasks, np.uint32, local_rank)
    d_run = DA(n_serv, np.uint32, local_rank)

    def step():
        # Returns after the batch's results are final in HBM (stream sync inside).
        if sharded:
            ctx.dispatch_sharded(d_env, d_minv, d_ip, d_out, None, d_run)
        else:
            ctx.dispatch_device(d_env, d_minv, d_ip, d_out, None, d_run)

    # One GPU: the steps are pipelined two deep (ydc_dispatch_device_async / ydc_dispatch_wait):
    # batch k + 1 is enqueued before the host looks at the outcome of batch k, each into its own
    # result buffers. Every step still places the whole batch and waits for its results inside
    # the timed region; what disappears is the device idling while the host turns around.
    # (--no-pipeline: one synchronous call per step, as in rounds 1 and 2.)
    pipelined = not sharded and not use_dist and not args.no_pipeline and args.steps >= 2
    d_out2 = DA(n_tasks, np.uint32, local_rank) if pipelined else None
    d_run2 = DA(n_serv, np.uint32, local_rank) if pipelined else None

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

    run_steps(args.warmup)
    ctx.synchronize()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-batch latency (p50 / p99): synchronous calls, one batch in flight.
    lat = []
    for _ in range(min(args.steps, 1000) if pipelined else 0):
        s0 = time.perf_counter()
        step()
        lat.append(time.perf_counter() - s0)
    if pipelined:
        step()  # (d_out / d_run hold a synchronous batch's results for the checks below)
    sync_ms = 1e3 * sum(lat) / len(lat) if lat else None
    if not pipelined:
        # (N > 1: the timed steps are synchronous; their own latencies serve)
        barrier()
        for _ in range(min(args.steps, 100)):
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
    # p50 / p99 want >= 100 samples whatever --steps is (extra batches, outside the timed region).
    lat_all = list(lat)
    while len(lat_all) < 100:
        barrier()
        s0 = time.perf_counter()
        step()
        lat_all.append(time.perf_counter() - s0)

    # SURVEY.md §8(d)'s assignments/s: batch visible to the dispatcher -> all results visible to
    # the host, through the host-buffer entry point ydc_dispatch (H2D of 12 B/request over
    # PCIe, kernels, D2H of 4 B/request). Reported in `end_to_end`, next to `value`.
    e2e = None
    if not use_dist and not args.resident_only:
        # (columns and the result array are the caller's and stay the same from batch to batch,
        # as in a scheduler loop; allocating them per call would time numpy, not the dispatch)
        # The scheduler's buffers are page-locked once (ydc_host_alloc): ydc_dispatch then reads
        # the columns and writes the placement in place — no staging copy on either side. The
        # same call with pageable numpy arrays (staged through the library's own pinned arenas)
        # is reported beside it.
        def host_loop(cols, res):
            ctx.dispatch(cols, want_util=False, want_running=False, out_idx=res)
            hl = []
            for _ in range(max(100, min(1000, args.steps))):
                s0 = time.perf_counter()
                ctx.dispatch(cols, want_util=False, want_running=False, out_idx=res)
                hl.append(time.perf_counter() - s0)
            return hl, {"assignments_per_s": st["granted"] * len(hl) / sum(hl),
                        "ms_per_batch": 1e3 * sum(hl) / len(hl),
                        "p50_ms": 1e3 * percentile(hl, 0.50), "p99_ms": 1e3 * percentile(hl, 0.99),
                        "batches": len(hl)}

        tk_c = {k: np.ascontiguousarray(v, dtype=np.uint32) for k, v in tk.items()}
        res = np.empty(len(tk_c["env_id"]), np.uint32)
        _, pageable = host_loop(tk_c, res)
        tk_p = {k: binding.pinned_empty(len(v), np.uint32) for k, v in tk_c.items()}
        for k in tk_c:
            tk_p[k][:] = tk_c[k]
        res_p = binding.pinned_empty(len(res), np.uint32)
        _, e2e = host_loop(tk_p, res_p)
        e2e_same = bool(np.array_equal(res_p, res))
        e2e["definition"] = ("ydc_dispatch with the caller's page-locked host buffers (ydc_host_alloc), "
                             "batch visible to the dispatcher -> placement visible to the host: the "
                             "kernels read the columns and write the results over PCIe in place, "
                             "SURVEY.md 8(d)")
        e2e["same_placement_as_pageable"] = e2e_same
        pageable["definition"] = ("the same call with pageable numpy arrays: staged through the "
                                  "library's pinned arenas (two host memcpys + two copy commands)")
        e2e["pageable_buffers"] = pageable

    # N > 1: the placement of the whole batch against the oracle (one more step, gathered).
    parity_oracle = None
    if use_dist and dist:
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
    n_prof = max(3, min(40, args.steps))
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

    if rank == 0:
        host_idx = d_out.numpy()
        shape = "%d pending requests x %d servants" % (n_all, n_serv)
        out = {
            "metric": METRIC,
            "value": granted_all * args.steps / elapsed,
            "unit": "assignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
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
                           args.config, " with %d digests" % args.digests if args.digests else "",
                           "" if world == 1 else " (%s scaling)" % args.scaling, shape,
                           ", %.0f %% of the servants on shared hosts" % (100 * args.shared_ip_frac)
                           if args.shared_ip_frac else "", st["n_classes"]),
                       "parallelism": "1 GPU" if world == 1 else
                                      ("one global batch of %d requests x %d servants sharded by "
                                       "rank range over %d GPUs, %s all-gather of boundary states "
                                       "and servant-slot deltas" % (
                                           n_all, n_serv, world,
                                           "RCCL" if transport == "rccl" else "mailbox (%s)" % transport)
                                       if sharded else group_note),
                       "inputs": "request columns + servant table resident in HBM; results in HBM"},
            "p99_dispatch_latency_ms": 1e3 * percentile(lat_all, 0.99),
            "p50_dispatch_latency_ms": 1e3 * percentile(lat_all, 0.50),
            "latency_samples": len(lat_all),
            "stats": {k: v for k, v in st.items() if k != "stage_ms"},
            "stage_ms": stage_ms,
            "kernels_us_per_step": {k: 1e3 * v[1] / n_prof for k, v in prof.items()},
            "kernel_launches_per_step": {k: v[0] / n_prof for k, v in prof.items()},
        }
        if e2e:
            out["end_to_end"] = e2e
            out["host_buffers_assignments_per_s"] = e2e["assignments_per_s"]
        if use_dist:
            # sharded: the N ranks placed ONE global batch through ydc_dispatch_sharded (false:
            # no transport came up and every rank placed its own batch — not a scaling result).
            out["sharded"] = bool(sharded)
            out["transport"] = {"ipc-host": "ipc"}.get(transport, transport)
            out["transport_detail"] = transport + ("" if not group_note else " (%s)" % group_note)
            out["rccl_ranks"] = rccl_ranks
            out["rccl"] = is_rccl
            out["parity_vs_oracle"] = parity_oracle
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
                               "traffic": pmc_traffic(dom, args.config),
                               "algorithmic_bytes_per_launch": alg_bytes,
                               "avg_launch_us": avg_launch_s * 1e6,
                               "launches_per_step": launches / n_prof,
                               # The same bytes against the whole step (all kernels + host
                               # turn-around): what a batch achieves, whatever the launch count.
                               "per_step_GBps": alg_bytes / (elapsed / args.steps) / 1e9,
                               "note": "k_match_pass: matching passes 0 + 1 are one launch on one "
                                       "GPU (two until round 2: half the duration per launch, the "
                                       "same per batch); dependency-bound, see DESIGN.md 3.4"}
        if world == 1 and not args.no_cpu_baseline:
            ref_idx, base = cpu_baseline(sv, tk, max_tasks=100_000)
            out["cpu_baseline"] = base
            out["parity_vs_cpu_baseline"] = bool(np.array_equal(ref_idx, host_idx[:len(ref_idx)]))
        line = json.dumps(out)
    else:
        line = None
    stuck = init_thread is not None and init_thread.is_alive()  # still inside ncclCommInitRank
    if sharded:
        if dist:
            dist.barrier()  # nobody unmaps a mailbox a peer may still be writing to
        ctx.group_destroy()
    ctx.close()
    # ONE JSON line, and the last thing on stdout: RCCL prints a version banner through C
    # stdio, which would otherwise be flushed behind it at exit. Every rank flushes before the
    # last barrier; rank 0 prints after it.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if line:
        print(line, flush=True)
    if stuck:
        os._exit(0)  # (do not tear the context down under a bootstrap that never returned)


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
    main()
