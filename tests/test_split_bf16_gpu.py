"""The opt-in split-bf16 GEMM (JIRF_GEMM_SPLIT_BF16=1) must be as accurate as the fp32-MFMA GEMM.

Both kernels are run in child processes (the switch is read once per process) on the same seeded
operands and compared with a float64 product; the split form keeps six of the nine bf16 partial products
of every fp32 pair and accumulates in fp32, so its error has to stay within a small factor of the fp32
kernel's own rounding error."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(split):
    path = os.pathsep.join([ROOT, os.path.join(ROOT, "image-restoration_amd"), os.environ.get("PYTHONPATH", "")])
    # JIRF_OP_LN_FOLD: the LayerNorm weight and bias folded into the GEMM weights, the form the networks run (the
    # split kernel only normalises)
    env = dict(os.environ, JIRF_GEMM_SPLIT_BF16="1" if split else "0", JIRF_OP_LN_FOLD="1", PYTHONPATH=path)
    r = subprocess.run([sys.executable, "-m", "tests.split_bf16_probe"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_split_bf16_matches_fp32_accuracy():
    fp32, split = _probe(False), _probe(True)
    assert fp32.keys() == split.keys()
    for k in fp32:
        # both are a few ulp of the output magnitude (|y| ~ 5); the absolute floor covers tiny fp32 errors
        assert split[k] <= 3.0 * fp32[k] + 2e-6, (k, fp32[k], split[k])
        assert split[k] < 5e-5, (k, split[k])
