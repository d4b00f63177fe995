"""The committed golden vectors (generated from the verbatim reference by
tests/golden/make_golden.py) against the oracle restatement and the CPU model."""
import glob
import os

import numpy as np
import pytest

from oracle import oraclebind as O
from tests.model import modelbind as M

ALL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
FILES = [f for f in ALL if "_prefix_" not in f and "_stream_" not in f]  # whole small batches (inputs inside)


def test_fixtures_exist():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_and_model_reproduce_reference_vectors(path):
    z = np.load(path)
    sv = {k[3:]: z[k] for k in z.files if k.startswith("sv_")}
    tk = {k[3:]: z[k] for k in z.files if k.startswith("tk_")}
    for method in ("scan", "sorted"):
        idx, _, run = O.dispatch(sv, tk, method)
        assert np.array_equal(idx, z["ref_servant_idx"]), method
        assert np.array_equal(run, z["ref_running_after"]), method
    idx, _, run, _ = M.dispatch(sv, tk, 128)
    assert np.array_equal(idx, z["ref_servant_idx"]) and np.array_equal(run, z["ref_running_after"])


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_full_size_pools_prefix_matches_reference(cfg):
    """BASELINE.json configs[2] / configs[3] at their real pool sizes (8k / 16k servants): the
    slot-order restatement used for the full-size GPU parity checks reproduces the verbatim
    reference on the first 50k requests of the batch."""
    from tests import cases
    from yadcc_amd import synth
    sv, tk = synth.make_config(cfg)
    ref = cases.reference_prefix(cfg, sv, tk)
    head = {k: v[:len(ref)] for k, v in tk.items()}
    idx, _, _ = O.dispatch(sv, head, "sorted")
    assert np.array_equal(idx, ref)


@pytest.mark.parametrize("cfg,n", [("cfg3", 400_000), ("cfg4", 200_000)])
def test_full_size_pools_long_prefix_digests(cfg, n):
    """... and on the long prefixes pinned block by block (tests/golden/ref_<cfg>_prefix_digests.npz:
    cfg3's first 400k requests — past the dedicated-tier boundary —, cfg4's first 200k)."""
    from tests import cases
    from yadcc_amd import synth
    sv, tk = synth.make_config(cfg)
    head = {k: v[:n] for k, v in tk.items()}
    idx, _, _ = O.dispatch(sv, head, "sorted")
    assert cases.check_prefix_digests(cfg, sv, tk, idx) == n


def test_stream_fixture_first_ticks_match_the_oracle():
    """tests/golden/ref_cfg5_stream_200_ticks.npz (the verbatim reference replaying configs[4]'s
    event stream): the oracle restatement, fed the same stream, reproduces the reference's
    digests on the first ticks — what the 200-tick GPU test relies on is the stream generator and
    the digest, both exercised here on CPU."""
    from yadcc_amd import streaming, synth
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_cfg5_stream_200_ticks.npz"))
    assert int(fx["ticks"]) >= 200
    sv, _ = synth.make_config("cfg5")
    es = streaming.EventStream(sv, 10_000, 10_000)
    for t in range(4):
        _, _, _, tk = es.next_tick()
        idx, _, run = O.dispatch(es.registry_snapshot(), tk, "sorted")
        assert synth.placement_hash(idx) == int(fx["digest"][t]), t
        assert int((idx < O.IDX_ENV_NOT_FOUND).sum()) == int(fx["granted"][t])
        es.commit(idx)
        assert synth.placement_hash(es.running.astype(np.uint32)) == int(fx["run_digest"][t])
        assert np.array_equal(run, es.running.astype(np.uint32))
