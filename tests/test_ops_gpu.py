"""Known-answer tests of the single-kernel entry points against the oracle's ops (GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import common as oc
from tests.gpu_util import call, dev, hptr, max_abs, ptr, stream

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)


def rnd(*shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


# fp32 MFMA and oneDNN/MKL sum in different orders: tolerance scales with sqrt(K)
def gemm_tol(k, mag=1.0):
    return 2e-6 * np.sqrt(k) * mag + 1e-6


@pytest.mark.parametrize("B,Cin,Cout,P", [
    (1, 48, 254, 96 * 64),      # level-1 project_in of HINT (odd Cout)
    (2, 127, 48, 1000),         # odd K, P not a multiple of the tile
    (1, 384, 1152, 32 * 32),    # latent qkv
    (3, 96, 96, 77),            # P not a multiple of 4 -> scalar loader
    (1, 1021, 384, 256),        # latent project_out (odd K)
    (1, 16, 3, 4096),           # tiny Cout
    (2, 192, 510, 128 * 128),   # big enough for the 128x128 tile
    (2, 255, 96, 512 * 256 + 24),   # weight-stationary streaming kernel: 3 row tiles, odd K, a ragged last pixel group
    (4, 127, 48, 256 * 256),        # ... 2 row tiles (the second half empty), odd K
    (5, 96, 24, 240 * 256),         # ... 1 row tile, groups of one wave running into the next image
    (1, 255, 96, 256 * 256),        # ... a single image: 1024 groups, one per SIMD, waves dealt across the workgroups
    (1, 192, 96, 2000 * 64 + 8),    # ... 2001 groups: two rounds of one per SIMD, a ragged last group
    (1, 384, 96, 488 * 328),        # 2501 groups: between the rounds, the K-loop kernel's 96 x 128 tile
])
def test_conv1x1(B, Cin, Cout, P):
    x, w, b = rnd(B, Cin, P), rnd(Cout, Cin, scale=Cin ** -0.5), rnd(Cout)
    res = rnd(B, Cout, P)
    ref = torch.einsum("oi,bip->bop", torch.from_numpy(w), torch.from_numpy(x)) + torch.from_numpy(b)[None, :, None]
    xd, rd = dev(x), dev(res)
    y = torch.empty(B, Cout, P, device="cuda")
    call("jirf_op_conv1x1", ptr(xd), hptr(w), hptr(b), None, ptr(y), B, Cin, Cout, P, stream())
    assert max_abs(y.cpu(), ref) < gemm_tol(Cin, 4)
    call("jirf_op_conv1x1", ptr(xd), hptr(w), None, ptr(rd), ptr(y), B, Cin, Cout, P, stream())
    assert max_abs(y.cpu(), ref - torch.from_numpy(b)[None, :, None] + torch.from_numpy(res)) < gemm_tol(Cin, 4)


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("B,C,Cout,P", [(2, 48, 144, 2048), (1, 384, 254, 130), (1, 96, 510, 4096)])
def test_ln_conv1x1(kind, B, C, Cout, P):
    x = rnd(B, C, P) * 2 + 0.5
    lw, lb = (1 + 0.1 * rnd(C)), 0.1 * rnd(C)
    w = rnd(Cout, C, scale=C ** -0.5)
    n = oc.channel_layernorm(torch.from_numpy(x).reshape(B, C, P, 1), torch.from_numpy(lw), torch.from_numpy(lb), kind == 1)
    ref = torch.einsum("oi,bip->bop", torch.from_numpy(w), n.reshape(B, C, P))
    y = torch.empty(B, Cout, P, device="cuda")
    xd, lwd, lbd = dev(x), dev(lw), dev(lb)
    call("jirf_op_ln_conv1x1", ptr(xd), ptr(lwd), ptr(lbd), kind, hptr(w), ptr(y), B, C, Cout, P, stream())
    assert max_abs(y.cpu(), ref) < gemm_tol(C, 8)
    yl = torch.empty(B, C, P, device="cuda")
    call("jirf_op_layernorm", ptr(xd), ptr(lwd), ptr(lbd), ptr(yl), B, C, P, kind, stream())
    assert max_abs(yl.cpu(), n.reshape(B, C, P)) < 2e-5


CONV3_SHAPES = [(1, 3, 48, 40, 56), (2, 48, 24, 32, 48), (1, 96, 192, 16, 24), (1, 96, 3, 24, 40), (1, 384, 768, 6, 10),
                (2, 16, 3, 37, 44), (1, 8, 4, 9, 260), (1, 8, 1, 5, 8), (1, 8, 2, 7, 30)]
# pixel shuffle needs Cout % 4 == 0, pixel unshuffle even H and W: only the combinations that exist are generated
CONV3_CASES = [(m,) + sh for m in (0, 1, 2) for sh in CONV3_SHAPES
               if not (m == 2 and sh[2] % 4) and not (m == 1 and (sh[3] % 2 or sh[4] % 2))]


@pytest.mark.parametrize("mode,B,Cin,Cout,H,W", CONV3_CASES)
def test_conv3x3(mode, B, Cin, Cout, H, W):
    x, w = rnd(B, Cin, H, W), rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1)
    res = None
    if mode == 1:
        ref = oc.pixel_unshuffle2(ref)
    elif mode == 2:
        ref = oc.pixel_shuffle2(ref)
    else:
        res = rnd(B, Cout, H, W)
        ref = ref + torch.from_numpy(res)
    y = torch.empty(ref.shape, device="cuda")
    xd = dev(x)
    rd = None if res is None else dev(res)
    call("jirf_op_conv3x3", ptr(xd), hptr(w), None, ptr(rd), ptr(y), B, Cin, Cout, H, W, mode, stream())
    assert max_abs(y.cpu(), ref) < gemm_tol(9 * Cin, 4)


def test_pixel_shuffle_bit_exact():
    """(Un)shuffle is pure indexing: a conv with an identity centre tap must reproduce it bit for bit."""
    B, C, H, W = 1, 8, 16, 24
    x = rnd(B, C, H, W)
    w = np.zeros((C, C, 3, 3), np.float32)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    xd = dev(x)
    for mode, fn in ((1, oc.pixel_unshuffle2), (2, oc.pixel_shuffle2)):
        ref = fn(torch.from_numpy(x))
        y = torch.empty(ref.shape, device="cuda")
        call("jirf_op_conv3x3", ptr(xd), hptr(w), None, None, ptr(y), B, C, C, H, W, mode, stream())
        assert np.array_equal(y.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("gate", [0, 1, 2])
@pytest.mark.parametrize("B,C,H,W,bias", [(1, 254, 40, 56, False), (2, 30, 17, 70, True), (1, 144, 64, 64, False),
                                          (1, 6, 5, 3, True), (1, 24, 82, 122, True)])
def test_dwconv3x3(gate, B, C, H, W, bias):
    x, w = rnd(B, C, H, W), rnd(C, 1, 3, 3, scale=1 / 3)
    b = rnd(C) if bias else None
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None if b is None else torch.from_numpy(b), padding=1, groups=C)
    if gate == 1:
        x1, x2 = ref.chunk(2, dim=1)
        ref = oc.gelu(x1) * x2
    elif gate == 2:     # ASTv2 FeedForward: only the first half is convolved
        h = C // 2
        bb = None if b is None else torch.from_numpy(b[:h])
        ref = F.conv2d(torch.from_numpy(x[:, :h]), torch.from_numpy(w[:h]), bb, padding=1, groups=h) * torch.from_numpy(x[:, h:])
    y = torch.empty(ref.shape, device="cuda")
    xd, wd = dev(x), dev(w)
    bd = None if b is None else dev(b)
    call("jirf_op_dwconv3x3", ptr(xd), ptr(wd), ptr(bd), ptr(y), B, C, H, W, gate, stream())
    assert max_abs(y.cpu(), ref) < 5e-6


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("C", [20, 32, 48, 64, 96, 100])
def test_layernorm_large_plane(kind, C):
    """Planes of >= 256k pixels take the float4 kernels: the register-resident single-pass form for C <= 96
    (three channel-count variants) and the three-pass form above it."""
    B, P = 2, 131072 + 256
    x = (RNG.standard_normal((B, C, P)) * 2 + 0.5).astype(np.float32)
    lw, lb = (1 + 0.1 * rnd(C)), 0.1 * rnd(C)
    n = oc.channel_layernorm(torch.from_numpy(x).reshape(B, C, P, 1), torch.from_numpy(lw), torch.from_numpy(lb), kind == 1)
    xd, lwd, lbd = dev(x), dev(lw), dev(lb)
    y = torch.empty(B, C, P, device="cuda")
    call("jirf_op_layernorm", ptr(xd), ptr(lwd), ptr(lbd), ptr(y), B, C, P, kind, stream())
    assert max_abs(y.cpu(), n.reshape(B, C, P)) < 2e-5


def test_gelu_over_every_float():
    """The kernels evaluate nn.GELU's erf form through one branch-free polynomial + power of two for erfc.  Scanned on
    the device over EVERY float in [-8.5, 8.5] (that is |x| <= 6 for erf(x) itself, x = v / sqrt 2) against float64:
    the erfc value within 2e-7 absolute, gelu within 5e-7 absolute (for v >= 0 within 2.5 spacings of the exact
    value's float; for v < 0 gelu decays to zero and only the absolute error means anything), and within 1e-6 of the
    form on the C library's erff that it replaced.  A second scan up to 1e30 checks the tail (gelu(v) -> max(v, 0))."""
    out = (C.c_double * 4)()
    call("jirf_op_gelu_error_scan", 8.5, out, stream())
    erfc_err, gelu_err, gelu_ulp, vs_erff = (float(v) for v in out)
    print(f"[gelu scan, every float in [-8.5, 8.5]] max |erfc err| = {erfc_err:.3e}, max |gelu err| = {gelu_err:.3e} "
          f"({gelu_ulp:.2f} spacings), max |gelu - erff form| = {vs_erff:.3e}")
    assert erfc_err <= 2e-7 and gelu_ulp <= 2.5 and gelu_err <= 1e-6 and vs_erff <= 1e-6
    call("jirf_op_gelu_error_scan", 1e30, out, stream())
    assert float(out[0]) <= 2e-7 and float(out[2]) <= 2.5 and np.isfinite(float(out[1]))
