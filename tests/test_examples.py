"""The scripts under examples/ run as a user would run them (GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,arg,expect", [("batch_qp.py", "256", "solved: 256 of 256"), ("facade_general_cost.py", "64", "dynamics violation")])
def test_example_runs(built, script, arg, expect):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), arg], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and expect in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
