"""
Short runs of the randomised differential tools (tools/fuzz_*.py) as part of the GPU suite: planned vs direct
interpolation kernel, bucket-grid KNN vs the oracle's brute force, ExportData end to end vs the oracle, refine() with the
HIP kernels vs the same host logic on the oracle kernels.  Each tool runs once, in its own process, one after the other.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seed,cases", [("fuzz_interp.py", 21, 60), ("fuzz_knn.py", 22, 60), ("fuzz_export.py", 23, 30),
                                             ("fuzz_refine_gpu.py", 24, 12)])
def test_fuzz_tool(tool, seed, cases):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), str(cases)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    tail = "\n".join((run.stdout + run.stderr).splitlines()[-15:])
    assert run.returncode == 0 and f"{cases} cases, 0 mismatches" in run.stdout, tail
