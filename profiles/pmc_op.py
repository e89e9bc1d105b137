"""Run ONE launch shape of the LayerNorm + 1x1 GEMM (the dominant kernel of the HINT benchmark)
a few times, so that a `rocprofv3 --pmc` pass over this program attributes its counters to that
shape only.  Usage: python3 profiles/pmc_op.py [B Cin Cout P] (default 32 96 510 65536)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-restoration_amd"))
import jirf_hip  # noqa: E402
import ctypes  # noqa: E402

B, C, M, P = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 96, 510, 65536)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, C, P), dtype=np.float32)).cuda()
lw = torch.ones(C, device="cuda")
lb = torch.zeros(C, device="cuda")
w = np.ascontiguousarray(rng.standard_normal((M, C), dtype=np.float32) * 0.1)
y = torch.empty(B, M, P, device="cuda")
fp = ctypes.POINTER(ctypes.c_float)


def dptr(t):
    return ctypes.cast(t.data_ptr(), fp)


lib = jirf_hip.lib()
for _ in range(4):
    rc = lib.jirf_op_ln_conv1x1(dptr(x), dptr(lw), dptr(lb), 1, w.ctypes.data_as(fp), dptr(y), B, C, M, P, None)
    assert rc == 0, jirf_hip.last_error()
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
