"""The multi-GPU sharding protocol (product code: ydc_dispatch_sharded) run by
REAL processes over gloo, world_size 2 and 3, on the CPU: every rank replays its slice with the
shared placement code compiled for the host (tests/model), the three exchanges of the protocol
are torch.distributed all-gathers, and rank 0 checks the concatenated placement and the summed
slot deltas against the oracle. No GPU involved."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from tests.model import modelbind
from yadcc_amd import pack


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, case_kw, cuts, chunk, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sv, tk = cases.random_case(**{k: v for k, v in case_kw.items() if k != "oracle"})
        a = pack.to_abi_columns(sv)
        lo, hi = cuts[rank], cuts[rank + 1]
        t = {k: np.ascontiguousarray(tk[k][lo:hi], dtype=np.uint32)
             for k in ("env_id", "min_version", "requestor_ip")}
        L = modelbind.lib()
        L.model_shard_open.restype = C.c_void_p
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        S, N = len(a["version"]), hi - lo
        h = L.model_shard_open(
            C.c_uint32(S), p(a["version"]), p(a["num_processors"]), p(a["current_load"]),
            p(a["max_tasks"]), p(a["running_tasks"]), p(a["flags"]), p(a["env_mask"]),
            p(a["ip_id"]), C.c_uint32(N), p(t["env_id"]), p(t["min_version"]),
            p(t["requestor_ip"]), C.c_uint32(chunk))
        assert h, "model_shard_open failed"
        h = C.c_void_p(h)
        nc = L.model_shard_n_classes(h)
        # (1) consuming-request counts of the slices, one per independent part of the registry
        #     -> base of this rank's level guesses
        G = L.model_shard_n_parts(h)
        cnt = np.zeros(G, np.uint32)
        L.model_shard_consuming_parts(h, p(cnt))
        totals = [torch.zeros(G, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(totals, torch.from_numpy(cnt.astype(np.int64)))
        base = np.zeros(G, np.uint32)
        for x in totals[:rank]:
            base += x.numpy().astype(np.uint32)
        # (2) passes: publish (end state of the last chunk, busy count), stop when nobody is busy
        bounds = None
        passes = 0
        while True:
            out_end = np.zeros(nc * 4, dtype=np.uint32)
            # the nearest rank below that has requests (ranks without requests are transparent)
            pred = max([q for q in range(rank) if cuts[q + 1] > cuts[q]], default=None)
            bin_ = None if (pred is None or bounds is None) else p(bounds[pred])
            busy = L.model_shard_pass(h, C.c_uint32(passes), p(base), bin_, p(out_end))
            rec = torch.from_numpy(np.concatenate([out_end, np.array([busy], np.uint32)]).astype(np.int64))
            got = [torch.zeros_like(rec) for _ in range(world)]
            dist.all_gather(got, rec)
            bounds = [np.ascontiguousarray(g.numpy()[:-1].astype(np.uint32)) for g in got]
            passes += 1
            if sum(int(g[-1]) for g in got) == 0:
                break
            assert passes < 10_000
        # (3) placement of the slice + all-gather of the per-servant slot deltas
        idx = np.empty(N, np.uint32)
        delta = np.zeros(S, np.uint32)
        L.model_shard_finalize(h, p(idx), p(delta))
        dl = [torch.zeros(S, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(dl, torch.from_numpy(delta.astype(np.int64)))
        running = a["running_tasks"].astype(np.int64) + sum(d.numpy() for d in dl)
        # collect the placement on rank 0
        sizes = [cuts[r + 1] - cuts[r] for r in range(world)]
        pad = max(sizes + [1])
        mine = torch.full((pad,), -1, dtype=torch.int64)
        mine[:N] = torch.from_numpy(idx.astype(np.int64))
        allp = [torch.zeros(pad, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allp, mine)
        L.model_shard_close(h)
        if rank == 0:
            want, _, wrun = O.dispatch(sv, tk, case_kw.get("oracle", "sorted"))
            got = np.concatenate([allp[r].numpy()[:sizes[r]] for r in range(world)]).astype(np.uint32)
            ret["placement_ok"] = bool(np.array_equal(got, want))
            ret["running_ok"] = bool(np.array_equal(running.astype(np.uint32), wrun))
            ret["passes"] = passes
            ret["parts"] = int(G)
            ret["timeouts"] = int((want == O.IDX_TIMEOUT).sum())
    finally:
        dist.destroy_process_group()


def _run(world, case_kw, cuts, chunk=256):
    # torch only inside the test (and its children): the GPU test process of this repo must not
    # load a second HIP runtime by merely collecting this file.
    import torch.multiprocessing as mp
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_rank_main, args=(world, _free_port(), case_kw, cuts, chunk, ret), nprocs=world,
             join=True)
    assert ret.get("placement_ok") is True, dict(ret)
    assert ret.get("running_ok") is True, dict(ret)
    return dict(ret)


def test_two_ranks_match_oracle():
    kw = dict(seed=71, n_tasks=12_000, n_servants=400, n_envs=4, self_frac=0.15,
              unknown_env_frac=0.002)
    r = _run(2, kw, [0, 5000, 12_000])
    assert r["passes"] >= 2


def test_three_ranks_with_an_empty_slice_and_timeouts():
    kw = dict(seed=72, n_tasks=9000, n_servants=120, n_envs=2, oversubscribed=True)
    r = _run(3, kw, [0, 4000, 4000, 9000], chunk=128)
    assert r["timeouts"] > 100


def test_two_ranks_hosts_with_several_servants():
    """`self` for hosts that run several servants is resolved at replay time from the class
    state, so such registries shard like any other (checked against the literal restatement)."""
    kw = dict(seed=73, n_tasks=6000, n_servants=160, n_envs=3, shared_ip_frac=0.3, self_frac=0.4,
              oracle="scan")
    _run(2, kw, [0, 2500, 6000], chunk=128)


def test_two_ranks_disjoint_environment_partitions():
    """Disjoint environment partitions = independent parts of the registry, each consumed at
    its own requests' rate: the per-part level guesses are exact, the ranks agree after the
    first pass (the global level would need about one pass per chunk)."""
    kw = dict(seed=74, n_tasks=20_000, n_servants=400, n_envs=4, disjoint_envs=True, self_frac=0.0)
    r = _run(2, kw, [0, 9000, 20_000], chunk=128)
    assert r["parts"] == 4 and r["passes"] <= 3
