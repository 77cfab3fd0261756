"""GPU: the randomised parity runs (tests/fuzz_seg.py, tests/fuzz_reg.py -- the oracle as checker on random tiles, overlaps, ragged
volumes, widths, BN, batch sizes, tile ranges, network shapes and step trees) in a bounded form, as child processes.  They found what
the fixed-geometry tests did not twice: assemble's crop quirks (round 2) and a batch-dependent summation order (round 4)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,env", [("fuzz_seg.py", {"SEED": "0", "CASES": "4", "SHARED_CASES": "4"}),
                                        ("fuzz_seg.py", {"SEED": "7", "CASES": "2", "SHARED_CASES": "4"}),
                                        ("fuzz_seg.py", {"SEED": "3", "CASES": "4", "PRECISION": "f32"}),      # the exact-fp32 path (conv3_wino_f32): oracle parity + batch independence
                                        ("fuzz_reg.py", {})])
def test_randomised_parity(script, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)], capture_output=True, text=True, timeout=1500, cwd=ROOT,
                       env=dict(os.environ, **env))
    print(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
