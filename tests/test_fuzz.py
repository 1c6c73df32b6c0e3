"""A short run of the randomised closed-form parity sweep (tools/fuzz_parity.py): sizes drawn around
every switch point of the kernels, every block size, both dtypes; mahal_and_det, decompose, solve
and det against the planted solution and the closed-form log-determinant."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_closed_form_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seconds", "20", "--seed", "7"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK:" in r.stdout
