"""A bounded run of the randomised differential (tests/tools/fuzz_parity.py) inside `pytest -m gpu`:
200 seeded shapes — 1..8 digests, shared hosts, oversubscription, initial running_tasks, unknown
digests, self requests, forced chunk and ring sizes, batches committed in two halves — through
the HIP path and the oracle, every placement, utilisation and running_tasks column compared
bit for bit. The same seeds every run; the open-ended version is the developer tool itself."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("first_seed", [1000, 52000])
def test_bounded_fuzz_against_the_oracle(first_seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_parity.py"), "240",
                          str(first_seed), "100"], cwd=ROOT, capture_output=True, text=True, timeout=400)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert "fuzz: 100 cases, 0 mismatches" in out.stdout, tail
