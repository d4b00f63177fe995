"""Multi-GPU sharding protocol on the single GPU of the test box.

(1) ydc_group_init + ydc_dispatch_sharded over RCCL with a 1-rank communicator (the library
    loads librccl, creates the communicator, runs the three all-gathers).
(2) The real protocol with 2, 3 and 5 ranks: several contexts on device 0 as the ranks,
    exchanging through the single-process transport (ydc_group_init_local), each rank driven
    by its own thread. The concatenation of the ranks' results must equal the placement of the
    whole batch by the oracle, and every rank must end with the same running_tasks."""
import threading

import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from yadcc_amd import binding, pack, synth

pytestmark = pytest.mark.gpu
DA = binding.DeviceArray


def sharded_run(ctxs, sv, tk, cuts, commit=False):
    """cuts: slice boundaries [0, ..., N]. Returns (per-rank idx, per-rank util, per-rank running)."""
    G = len(ctxs)
    res = [None] * G
    errs = []

    def worker(r):
        try:
            c = ctxs[r]
            lo, hi = cuts[r], cuts[r + 1]
            d = [DA.from_numpy(tk[k][lo:hi]) for k in ("env_id", "min_version", "requestor_ip")]
            out = DA(hi - lo, np.uint32)
            util = DA(hi - lo, np.float64)
            run = DA(len(sv["version"]), np.uint32)
            c.dispatch_sharded(d[0], d[1], d[2], out, util, run, commit=commit)
            res[r] = (out.numpy(), util.numpy(), run.numpy(), c.stats())
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    [t.start() for t in ths]
    [t.join(120) for t in ths]
    assert not errs, errs
    assert all(x is not None for x in res), "a rank did not finish"
    return res


def make_group(G, sv):
    ctxs = [binding.Context(device=0) for _ in range(G)]
    cols = pack.to_abi_columns(sv)
    for c in ctxs:
        c.upload_servants(cols)
    binding.group_init_local(ctxs)
    return ctxs


def check_against_oracle(res, sv, tk, method="sorted"):
    want, wutil, wrun = O.dispatch(sv, tk, method)
    got = np.concatenate([r[0] for r in res])
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at request %d (gpu %d oracle %d), %d total" % (
        bad[0], got[bad[0]], want[bad[0]], bad.size)
    assert np.array_equal(np.concatenate([r[1] for r in res]), wutil)
    for r in res:
        assert np.array_equal(r[2], wrun)
    assert sum(r[3]["granted"] for r in res) == int((want < O.IDX_ENV_NOT_FOUND).sum())


_RCCL_SCRIPT = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
from oracle import oraclebind as O
from tests import cases
from tests.test_sharded_gpu import sharded_run, check_against_oracle
from yadcc_amd import binding, pack
t0 = time.time()
sv, tk = cases.random_case(seed=41, n_tasks=30_000, n_servants=700, n_envs=3, self_frac=0.1,
                           unknown_env_frac=0.001)
ctx = binding.Context(device=0)
ctx.upload_servants(pack.to_abi_columns(sv))
t1 = time.time()
print("[phase] context + registry up after %%.1f s" %% (t1 - t0), flush=True)
uid = binding.group_unique_id()
t2 = time.time()
print("[phase] ncclGetUniqueId took %%.1f s" %% (t2 - t1), flush=True)
ctx.group_init(uid, 0, 1)
t3 = time.time()
print("[phase] ncclCommInitRank took %%.1f s" %% (t3 - t2), flush=True)
res = sharded_run([ctx], sv, tk, [0, len(tk["env_id"])])
t4 = time.time()
check_against_oracle(res, sv, tk)
ctx.group_destroy()
ctx.close()
print("RCCL-1-RANK-OK setup %%.1fs unique_id %%.1fs comm_init %%.1fs dispatch %%.1fs" %% (
    t1 - t0, t2 - t1, t3 - t2, t4 - t3))
"""


def test_rccl_single_rank_group():
    """In a child process with a time limit (the one bench.py gives the same call). RCCL runs
    with NCCL_DEBUG=INFO: whatever happens, the log says how far the bootstrap got — a box where
    the communicator does not come up is skipped WITH that diagnosis, a wrong result never is.
    The log also goes to gpurun_out/rccl_1rank_debug.log (copied to profiles/ per round)."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG="INFO",
               NCCL_DEBUG_SUBSYS="INIT,BOOTSTRAP,NET,ENV", HSA_ENABLE_IPC_MODE_LEGACY="0")
    limit = float(os.environ.get("YDC_BENCH_RCCL_TIMEOUT", "240"))
    t0 = time.time()
    proc = subprocess.Popen([sys.executable, "-c", _RCCL_SCRIPT % root], env=env, cwd=root,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        so, _ = proc.communicate(timeout=limit)
        timed_out = False
    except subprocess.TimeoutExpired:
        proc.kill()
        so, _ = proc.communicate()
        timed_out = True
    took = time.time() - t0
    log_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(log_dir):
        with open(os.path.join(log_dir, "rccl_1rank_debug.log"), "w") as f:
            f.write("# child %s after %.1f s (limit %.0f s), rc %s\n" % (
                "KILLED" if timed_out else "ended", took, limit, proc.returncode))
            f.write(so or "")
    tail = "\n".join((so or "").strip().splitlines()[-25:])
    if timed_out:
        pytest.skip("RCCL 1-rank communicator did not come up within %.0f s on this box; "
                    "last lines of the NCCL_DEBUG=INFO log:\n%s" % (limit, tail))
    assert "RCCL-1-RANK-OK" in (so or ""), tail
    print([ln for ln in so.splitlines() if "RCCL-1-RANK-OK" in ln][-1], "(child %.1f s)" % took)


@pytest.mark.parametrize("G", [2, 3, 5])
def test_local_ranks_match_oracle(G, monkeypatch):
    """(G == 3: key windows + radix sort; otherwise the registry is small enough for every rank
    to bin-sort it whole, which the group path prefers.)"""
    if G == 3:
        monkeypatch.setenv("YDC_GROUP_BINSORT", "0")
    sv, tk = cases.random_case(seed=50 + G, n_tasks=60_000, n_servants=1200, n_envs=4,
                               self_frac=0.15, unknown_env_frac=0.002)
    n = len(tk["env_id"])
    ctxs = make_group(G, sv)
    cuts = [0] + sorted(np.random.default_rng(G).integers(0, n, G - 1).tolist()) + [n]
    res = sharded_run(ctxs, sv, tk, cuts)
    check_against_oracle(res, sv, tk)
    assert max(r[3]["rounds"] for r in res) == min(r[3]["rounds"] for r in res)  # lockstep
    if G == 3:
        assert all(r[3]["shard_sort_batches"] >= 1 for r in res)  # every rank sorted a key window
    else:
        assert all(r[3]["radix_passes"] == 0 for r in res)  # 0: the bin sort placed the slots
    [c.close() for c in ctxs]


def test_local_ranks_empty_slices_and_oversubscription():
    """Ranks without requests pass their predecessor's state on; the tail times out."""
    sv, tk = cases.random_case(seed=61, n_tasks=40_000, n_servants=300, n_envs=2,
                               oversubscribed=True)
    n = len(tk["env_id"])
    ctxs = make_group(4, sv)
    res = sharded_run(ctxs, sv, tk, [0, 0, n // 3, n // 3, n])  # ranks 0 and 2 are empty
    check_against_oracle(res, sv, tk)
    assert (np.concatenate([r[0] for r in res]) == binding.IDX_TIMEOUT).sum() > 1000
    [c.close() for c in ctxs]


def test_local_ranks_commit_two_batches():
    """COMMIT applies the global deltas on every rank: two sharded batches == one big batch."""
    sv, tk = cases.random_case(seed=62, n_tasks=50_000, n_servants=900, n_envs=3, self_frac=0.1)
    half = 25_000
    a = {k: v[:half] for k, v in tk.items()}
    b = {k: v[half:] for k, v in tk.items()}
    ctxs = make_group(2, sv)
    r1 = sharded_run(ctxs, sv, a, [0, 10_000, half], commit=True)
    r2 = sharded_run(ctxs, sv, b, [0, 20_000, half], commit=True)
    want, _, wrun = O.dispatch(sv, tk, "sorted")
    got = np.concatenate([r[0] for r in r1] + [r[0] for r in r2])
    assert np.array_equal(got, want)
    for c in ctxs:
        assert np.array_equal(c.get_running(), wrun)
    [c.close() for c in ctxs]


@pytest.mark.parametrize("n_tasks", [400_000, 4_000_000])
def test_cfg4_shape_eight_ranks(n_tasks):
    """BASELINE.json configs[3]: 4M requests x 16k servants, 4 digests, sharded over 8 ranks
    (here: 8 contexts on the one GPU of the test box), and a tenth of it."""
    sv, tk = synth.make_config("cfg4", n_tasks=n_tasks)
    n = len(tk["env_id"])
    ctxs = make_group(8, sv)
    cuts = [n * r // 8 for r in range(9)]
    res = sharded_run(ctxs, sv, tk, cuts)
    check_against_oracle(res, sv, tk)
    if n_tasks == 4_000_000:
        # the real configs[3] pool: pinned to the verbatim reference on the first 50k requests
        ref = cases.reference_prefix("cfg4", sv, tk)
        assert np.array_equal(np.concatenate([r[0] for r in res])[:len(ref)], ref)
    [c.close() for c in ctxs]


def test_sort_is_sharded_and_a_missed_window_falls_back(monkeypatch):
    """every rank generates and sorts only the key window its rank range can
    reach (stats: shard_sort_batches). A window that turns out too small — here forced with a
    margin of zero slots on a registry whose classes run far from the global level — is
    detected on every rank alike and the batch is repeated with the full sort: same results."""
    monkeypatch.setenv("YDC_GROUP_BINSORT", "0")  # (a registry this small is bin-sorted whole otherwise)
    sv, tk = cases.random_case(seed=66, n_tasks=120_000, n_servants=2500, n_envs=4,
                               unknown_env_frac=0.001, self_frac=0.1)
    n = len(tk["env_id"])
    cuts = [n * r // 4 for r in range(5)]
    ctxs = make_group(4, sv)
    res = sharded_run(ctxs, sv, tk, cuts)
    check_against_oracle(res, sv, tk)
    st = [r[3] for r in res]
    assert all(x["shard_sort_batches"] == 1 and x["shard_sort_misses"] == 0 for x in st), st
    assert all(x["n_slots"] < 0.7 * sum(y["n_slots"] for y in st) for x in st)  # windows, not all
    [c.close() for c in ctxs]
    # the same batch with windows that cannot hold the deviation of the classes
    monkeypatch.setenv("YDC_SHARD_MARGIN", "0")
    sv2, tk2 = cases.random_case(seed=67, n_tasks=60_000, n_servants=1500, n_envs=4,
                                 oversubscribed=True, self_frac=0.1)
    n2 = len(tk2["env_id"])
    ctxs = make_group(3, sv2)
    res = sharded_run(ctxs, sv2, tk2, [0, n2 // 3, 2 * n2 // 3, n2])
    check_against_oracle(res, sv2, tk2)
    st = [r[3] for r in res]
    assert len({(x["shard_sort_batches"], x["shard_sort_misses"]) for x in st}) == 1, st  # same verdict
    [c.close() for c in ctxs]
    monkeypatch.setenv("YDC_SHARD_SORT", "0")
    ctxs = make_group(3, sv2)
    res = sharded_run(ctxs, sv2, tk2, [0, n2 // 3, 2 * n2 // 3, n2])
    check_against_oracle(res, sv2, tk2)
    assert all(r[3]["shard_sort_batches"] == 0 for r in res)
    [c.close() for c in ctxs]


def test_late_change_behind_a_rank_without_requests():
    """configs[2]'s batch (its serial chain at the dedicated-tier boundary makes rank 0's end
    state change as late as pass 2-3) cut so that the rank behind rank 0 has no requests: the
    rank behind THAT one continues rank 0 directly (ranks without requests are transparent), so
    the late change reaches it in the very next pass and cannot slip past the "no rank changed
    anything" verdict."""
    sv, tk = synth.make_config("cfg3")
    n = len(tk["env_id"])
    ctxs = make_group(4, sv)
    res = sharded_run(ctxs, sv, tk, [0, 450_000, 450_000, 450_000, n])
    check_against_oracle(res, sv, tk)
    [c.close() for c in ctxs]


def test_local_ranks_many_classes():
    """> 64 classes (two classes per lane) across 3 ranks."""
    sv, tk = cases.random_case(seed=63, n_tasks=30_000, n_servants=1500, n_envs=7,
                               unknown_env_frac=0.002)
    n = len(tk["env_id"])
    ctxs = make_group(3, sv)
    res = sharded_run(ctxs, sv, tk, [0, n // 4, n // 2, n])
    check_against_oracle(res, sv, tk)
    assert res[0][3]["n_classes"] > 64
    [c.close() for c in ctxs]


@pytest.mark.parametrize("kind", ["shared_hosts", "many_classes"])
def test_local_ranks_unsharded_registries(kind):
    """Hosts with several servants (`self` resolved at replay time, sharded like any other
    registry) and a registry the sharded matching does not take (more than 256 classes: every
    rank places the whole batch redundantly and keeps its slice) — same results."""
    if kind == "shared_hosts":
        sv, tk = cases.random_case(seed=64, n_tasks=6000, n_servants=200, n_envs=3,
                                   shared_ip_frac=0.25, self_frac=0.3)
    else:
        sv, tk = cases.random_case(seed=65, n_tasks=9000, n_servants=3000, n_envs=10)
        sv["version"] = (20 + np.arange(3000) % 3).astype(np.uint32)
    n = len(tk["env_id"])
    ctxs = make_group(3, sv)
    res = sharded_run(ctxs, sv, tk, [0, n // 5, n // 5, n], commit=True)
    want, wutil, wrun = O.dispatch(sv, tk, "scan" if kind == "shared_hosts" else "sorted")
    assert np.array_equal(np.concatenate([r[0] for r in res]), want)
    assert np.array_equal(np.concatenate([r[1] for r in res]), wutil)
    for r, c in zip(res, ctxs):
        assert np.array_equal(r[2], wrun) and np.array_equal(c.get_running(), wrun)
    [c.close() for c in ctxs]
