"""Randomised stress (-m gpu): the drivers under tools/ generate workloads no hand-written case covers --
label-length laws, far-apart ids (escapes), singletons, very wide classes, huge counts, batch splits,
sub-batch sizes, host vs device batches, class-size laws that push every record size of the persistent EM loop past what a thread requests
ahead (tools/r6_shape_check.py) -- and compare the HIP path with the oracle: classes bit-exact,
alpha within 1e-9 after a fixed number of iterations, bias-corrected lengths within 1e-9.  A few seconds each here; run them longer by hand."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed", [("builder_stress.py", 21), ("em_stress.py", 22), ("bias_stress.py", 23), ("r6_shape_check.py", 0)])
def test_randomised_stress(built, gpu, script, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(seed), "12"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = "\n".join(r.stdout.strip().splitlines()[-3:])
    assert r.returncode == 0 and "all ok" in tail, tail + "\n" + r.stderr[-2000:]
