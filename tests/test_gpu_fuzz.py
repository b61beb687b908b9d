"""A fixed-seed slice of the randomised parity run (scripts/fuzz_parity.py): random tables,
plans, partitionings and execution routes, GPU against the CPU oracle."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_seed_fuzz_slice():
    # its own process: the run creates and frees many contexts' worth of tables
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "30", "7"],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout
