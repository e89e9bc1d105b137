"""Child process of test_split_bf16_gpu.py: runs the 1x1 GEMM entry points once and prints, as one JSON
line, their maximum absolute error against a float64 product.  Which GEMM kernel serves the calls is
decided by JIRF_GEMM_SPLIT_BF16 in the environment the parent passes, read once per process."""
import json
import sys

import numpy as np
import torch

from tests.gpu_util import call, dev, hptr, ptr, stream


def main():
    rng = np.random.default_rng(11)
    out = {}
    # reduction lengths the split kernel covers (32, 48, 64, 96; ragged pixel counts and output-row tails), and one it
    # does not (127: both runs then use the same fp32 kernel)
    for (B, C, Cout, P) in [(2, 48, 254, 4096), (1, 96, 510, 4100), (2, 96, 288, 777), (1, 64, 170, 1000), (1, 32, 96, 333),
                            (1, 127, 48, 1000),
                            # short outputs over a long reduction (the chunked form; it wants a grid that fills the chip)
                            (1, 255, 96, 65536), (1, 127, 48, 70001), (2, 192, 80, 32768),
                            # long reductions on a grid of >= 512 tiles of 128 x 128 (the K-loop form; ragged rows and pixels)
                            (4, 510, 200, 16380), (2, 192, 300, 32768)]:
        x = rng.standard_normal((B, C, P)).astype(np.float32) * 2 + 0.5
        w = (rng.standard_normal((Cout, C)) * C ** -0.5).astype(np.float32)
        res = rng.standard_normal((B, Cout, P)).astype(np.float32)
        xd, rd = dev(x), dev(res)
        y = torch.empty(B, Cout, P, device="cuda")
        call("jirf_op_conv1x1", ptr(xd), hptr(w), None, ptr(rd), ptr(y), B, C, Cout, P, stream())
        ref = np.einsum("oi,bip->bop", w.astype(np.float64), x.astype(np.float64)) + res
        out[f"conv1x1_{C}_{Cout}_{P}"] = float(np.abs(y.cpu().numpy() - ref).max())
        call("jirf_op_conv1x1", ptr(xd), hptr(w), None, None, ptr(y), B, C, Cout, P, stream())
        out[f"conv1x1_plain_{C}_{Cout}_{P}"] = float(np.abs(y.cpu().numpy() - (ref - res)).max())
        lw = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
        lb = (0.1 * rng.standard_normal(C)).astype(np.float32)
        x64 = x.astype(np.float64)
        mu = x64.mean(1, keepdims=True)
        n = (x64 - mu) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * lw[None, :, None] + lb[None, :, None]
        ref = np.einsum("oi,bip->bop", w.astype(np.float64), n)
        lwd, lbd = dev(lw), dev(lb)
        call("jirf_op_ln_conv1x1", ptr(xd), ptr(lwd), ptr(lbd), 1, hptr(w), ptr(y), B, C, Cout, P, stream())
        out[f"ln_conv1x1_{C}_{Cout}_{P}"] = float(np.abs(y.cpu().numpy() - ref).max())
    # dense 3x3 convolution as an implicit GEMM (plain, with residual, and followed by pixel-unshuffle)
    B, Cin, Cout, H, W = 2, 96, 192, 128, 128
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(bias).double(),
                                     padding=1).numpy()
    xd, rd = dev(x), dev(res)
    y = torch.empty(B, Cout, H, W, device="cuda")
    call("jirf_op_conv3x3", ptr(xd), hptr(w), hptr(bias), ptr(rd), ptr(y), B, Cin, Cout, H, W, 0, stream())
    out["conv3x3_res"] = float(np.abs(y.cpu().numpy() - (ref + res)).max())
    call("jirf_op_conv3x3", ptr(xd), hptr(w), hptr(bias), None, ptr(y), B, Cin, Cout, H, W, 1, stream())
    unsh = torch.nn.functional.pixel_unshuffle(torch.from_numpy(ref), 2).numpy()
    out["conv3x3_unshuffle"] = float(np.abs(y.cpu().numpy().reshape(unsh.shape) - unsh).max())
    # ... and with few outputs (the chunked form): first down-sampling convolution of HINT, 48 -> 24 + pixel-unshuffle
    B, Cin, Cout, H, W = 2, 48, 24, 256, 256
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, padding=1).numpy()
    xd = dev(x)
    y = torch.empty(B, Cout, H, W, device="cuda")
    call("jirf_op_conv3x3", ptr(xd), hptr(w), None, None, ptr(y), B, Cin, Cout, H, W, 0, stream())
    out["conv3x3_fewrows_plain"] = float(np.abs(y.cpu().numpy() - ref).max())
    call("jirf_op_conv3x3", ptr(xd), hptr(w), None, None, ptr(y), B, Cin, Cout, H, W, 1, stream())
    unsh = torch.nn.functional.pixel_unshuffle(torch.from_numpy(ref), 2).numpy()
    out["conv3x3_fewrows_unshuffle"] = float(np.abs(y.cpu().numpy().reshape(unsh.shape) - unsh).max())
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
