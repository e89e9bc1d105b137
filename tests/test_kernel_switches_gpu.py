"""The alternative kernels behind the environment switches (read once per process) must pass the same
golden cases as the default ones: each switch set runs tests/test_golden_gpu.py in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switches", [
    {"JIRF_WINATTN_VALU": "1", "JIRF_CROSSWIN_VALU": "1"},   # vector-ALU window attention kernels
    {"JIRF_GEMM_NO_XS": "1"},                                 # every GEMM on the K-loop kernel
    {"JIRF_GEMM_TILE": "3"},                                  # ... with 64 x 64 tiles, no few-output conv kernel
    {"JIRF_GEMM_TILE": "2"},                                  # ... 64 x 128 (the default only picks it at large batches)
    {"JIRF_GEMM_TILE": "1"},                                  # ... 128 x 128
    {"JIRF_GEMM_TILE": "0"},                                  # ... 96 x 128
    {"JIRF_ATTN_V_SPLIT": "1"},                               # attention * V on the split kernel for short outputs
    {"JIRF_GEMM_SPLIT_BF16": "0"},                            # every GEMM on the fp32 matrix instruction (split-bf16 forms off)
])
def test_golden_cases_with_switches(switches):
    env = dict(os.environ, **switches)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_golden_gpu.py", "-x", "-q", "--timeout", "100",
                        "--timeout-method", "thread"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=115)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
