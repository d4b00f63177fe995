"""The committed golden vectors (generated from the verbatim reference by
tests/golden/make_golden.py) against the oracle restatement and the CPU model."""
import glob
import os

import numpy as np
import pytest

from oracle import oraclebind as O
from tests.model import modelbind as M

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


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
