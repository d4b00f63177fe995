"""ydc_dispatch_sharded — the PRODUCT code — in 2 and 3 real processes on device 0, exchanging
through the mailbox transport (ydc_group_ipc_export / ydc_group_init_ipc: HIP IPC device
memory written by the peers' kernels, or the shared host segment), against the oracle.

What the in-process ranks of tests/test_sharded_gpu.py cannot show: per-process HIP runtimes
and streams, handles handed over a side channel, in-stream collectives between processes that
share one GPU (RCCL refuses two ranks on one device; this transport does not), and a rank that
reaches a collective long before its peers do."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import oraclebind as O
from tests import cases
from yadcc_amd import binding, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(G, spec, timeout=300, stagger=None):
    """Starts G rank processes (tests/mp_rank_worker.py); returns their result dicts."""
    with tempfile.TemporaryDirectory(prefix="ydc_mp_") as d:
        sp = os.path.join(d, "spec.json")
        json.dump(spec, open(sp, "w"))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
        procs = []
        for r in range(G):
            if stagger and r == stagger[0]:
                import time
                time.sleep(stagger[1])  # this rank shows up late: the others wait in-stream
            procs.append(subprocess.Popen([sys.executable, "-m", "tests.mp_rank_worker", d, str(r),
                                           str(G), sp], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=timeout))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        for r, (p, (so, se)) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and "RANK-%d-OK" % r in so, (r, so[-1500:], se[-3000:])
        return [dict(np.load(os.path.join(d, "result_%d.npz" % r), allow_pickle=False))
                for r in range(G)]


def check(res, sv, tk, method="sorted", batch=0, want=None):
    if want is None:
        want = O.dispatch(sv, tk, method)
    widx, wutil, wrun = want
    got = np.concatenate([r["idx_%d" % batch] for r in res])
    bad = np.nonzero(got != widx)[0]
    assert bad.size == 0, "first mismatch at request %d (gpu %d oracle %d), %d total" % (
        bad[0], got[bad[0]], widx[bad[0]], bad.size)
    assert np.array_equal(np.concatenate([r["util_%d" % batch] for r in res]), wutil)
    for r in res:
        assert np.array_equal(r["run_%d" % batch], wrun)
    st = [json.loads(str(r["stats_%d" % batch])) for r in res]
    assert sum(s["granted"] for s in st) == int((widx < O.IDX_ENV_NOT_FOUND).sum())
    assert len({s["rounds"] for s in st}) == 1  # lockstep
    return st


@pytest.mark.parametrize("G", [2, 3])
def test_processes_match_oracle(G):
    """General pool, 4 digests, self requests; G > 1 shards the slot sort as well."""
    kw = dict(seed=70 + G, n_tasks=60_000, n_servants=1200, n_envs=4, self_frac=0.15,
              unknown_env_frac=0.002)
    sv, tk = cases.random_case(**kw)
    n = len(tk["env_id"])
    cuts = [0] + sorted(np.random.default_rng(G).integers(0, n, G - 1).tolist()) + [n]
    # G == 2: every rank bin-sorts the (small) registry whole; G == 3: key windows + radix sort
    res = run_ranks(G, dict(case=kw, cuts=cuts, env={"YDC_GROUP_BINSORT": "0"} if G == 3 else {}))
    st = check(res, sv, tk)
    assert all(int(r["transport"]) in (binding.TRANSPORT_IPC_DEVICE, binding.TRANSPORT_IPC_HOST)
               for r in res)
    if G == 3:
        assert all(s["shard_sort_batches"] == 1 for s in st), st  # each rank sorted a key window only
    else:
        assert all(s["radix_passes"] == 0 for s in st), st
    print("transport:", binding.TRANSPORT_NAMES[int(res[0]["transport"])],
          "ms per rank:", [round(float(r["ms_0"]), 2) for r in res])


def test_processes_empty_slice_commit_two_batches_late_rank():
    """3 processes, the middle one without requests in the first batch, COMMIT, a second batch on
    top of the committed state — and rank 2 starts seconds after the others, which by then sit
    in their first in-stream exchange."""
    kw = dict(seed=81, n_tasks=50_000, n_servants=900, n_envs=3, self_frac=0.1)
    sv, tk = cases.random_case(**kw)
    half = 25_000
    res = run_ranks(3, dict(case=kw, batches=2, commit=True,
                            cuts=[[0, 9_000, 9_000, half], [0, 5_000, 20_000, half]]),
                    stagger=(2, 3.0))
    want, wutil, wrun = O.dispatch(sv, tk, "sorted")
    got = np.concatenate([r["idx_0"] for r in res] + [r["idx_1"] for r in res])
    assert np.array_equal(got, want)
    for r in res:
        assert np.array_equal(r["run_1"], wrun) and np.array_equal(r["resident_running"], wrun)


def test_processes_shared_hosts_host_segment():
    """Hosts that run several servants (`self` resolved at replay time; full sort on every rank),
    over the shared-host-segment flavour of the transport."""
    kw = dict(seed=64, n_tasks=6000, n_servants=200, n_envs=3, shared_ip_frac=0.25, self_frac=0.3)
    sv, tk = cases.random_case(**kw)
    n = len(tk["env_id"])
    res = run_ranks(2, dict(case=kw, cuts=[0, n // 3, n], transport="ipc-host"))
    check(res, sv, tk, method="scan")
    assert all(int(r["transport"]) == binding.TRANSPORT_IPC_HOST for r in res)


def test_processes_cfg4_tenth_four_ranks():
    """A tenth of BASELINE.json configs[3] (400k requests x 16k servants, 30 classes) over four
    processes: 64 KB of slot deltas per rank and exchange."""
    sv, tk = synth.make_config("cfg4", n_tasks=400_000)
    n = len(tk["env_id"])
    res = run_ranks(4, dict(config="cfg4", config_kw=dict(n_tasks=400_000),
                            cuts=[n * r // 4 for r in range(5)]), timeout=420)
    check(res, sv, tk)
