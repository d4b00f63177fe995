"""The committed golden vectors (generated from the verbatim reference by
tests/golden/make_golden.py) against the oracle restatement and the CPU model."""
import glob
import os

import numpy as np
import pytest

from oracle import oraclebind as O
from tests.model import modelbind as M

ALL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
FILES = [f for f in ALL if "_prefix_" not in f]  # whole small batches (inputs inside)


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
