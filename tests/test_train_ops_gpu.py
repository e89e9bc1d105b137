"""Backward / optimiser kernels of the training step against torch autograd over the oracle's operators
(torch-CPU fp32; the reference's Jittor autograd cannot run here, so this is "parity unpinned" like the
forward): gradients within 1e-3 relative of autograd's, the loss and the Adam update against a NumPy
restatement of Jittor's optimizer (jt.optim.Adam.step, Optimizer.clip_grad_norm).  The whole network's
backward composed from these operators is in test_train_step_gpu.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ast_oracle as so
from oracle import common as oc
from tests.gpu_util import call, dev, hptr, ptr, stream

pytestmark = pytest.mark.gpu
REL = 1e-3


def rnd(*shape, scale=1.0, seed=0):
    return (np.random.default_rng(seed + sum(shape)).standard_normal(shape) * scale).astype(np.float32)


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))


def test_smooth_l1_loss_and_gradient():
    n = 3 * 3 * 64 * 96
    pred, tgt = rnd(n, scale=1.5, seed=1), rnd(n, scale=1.5, seed=2)
    pt = torch.from_numpy(pred).requires_grad_(True)
    loss = F.smooth_l1_loss(pt, torch.from_numpy(tgt), reduction="none").mean() * 0.7
    loss.backward()
    g = torch.empty(n, device="cuda")
    out = C.c_double()
    pd_, td_ = dev(pred), dev(tgt)          # (held: a temporary's memory would be handed to the next allocation)
    on_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
    call("jirf_train_smooth_l1", ptr(pd_), ptr(td_), ptr(g), n, 0.7, ptr(on_dev), C.byref(out), stream())
    assert float(on_dev.item()) == out.value                 # the device copy of the loss (read later by the caller)
    assert abs(out.value - float(loss)) < 1e-6 * abs(float(loss))
    assert rel_err(g.cpu().numpy(), pt.grad.numpy()) < 1e-6
    assert (np.abs(pred - tgt) > 1).any() and (np.abs(pred - tgt) < 1).any()      # both branches


@pytest.mark.parametrize("torch_form,ranks", [(0, 1), (1, 1), (0, 4)])
def test_clip_and_adam_step(torch_form, ranks):
    """ranks > 1: the buffer holds the SUM over the ranks (what the all-reduce leaves) and grad_scale = 1 / ranks
    makes the update that of the mean, the clip's norm included."""
    n = 100003
    p, g = rnd(n, seed=3), rnd(n, scale=1e-3, seed=4)
    g_sum = (g * np.float32(ranks)).astype(np.float32)        # a power of two: g_sum / ranks == g exactly
    m, v = rnd(n, scale=1e-4, seed=5), np.abs(rnd(n, scale=1e-6, seed=6))
    lr, b0, b1, eps, step, max_norm = 2e-4, 0.9, 0.999, 1e-8, 7, 0.01
    # Jittor: Optimizer.clip_grad_norm(max_norm, 2) then Adam.step
    total = np.sqrt(np.sum(g.astype(np.float64) ** 2))
    gc = g * np.float32(min(max_norm / (total + 1e-6), 1.0))
    m2 = b0 * m + (1 - b0) * gc
    v2 = b1 * v + (1 - b1) * gc * gc
    if torch_form:
        p2 = p - lr / (1 - b0 ** step) * m2 / (np.sqrt(v2) / np.sqrt(1 - b1 ** step) + eps)
    else:
        p2 = p - m2 * (lr * np.sqrt(1 - b1 ** step) / (1 - b0 ** step)) / (np.sqrt(v2) + eps)
    pd, gd, md, vd = dev(p), dev(g_sum), dev(m), dev(v)
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    hs = C.c_double()
    call("jirf_train_sumsq", ptr(gd), n, ptr(ss), C.byref(hs), stream())
    assert abs(hs.value - (ranks * total) ** 2) < 1e-9 * (ranks * total) ** 2
    call("jirf_train_adam_step", ptr(pd), ptr(gd), ptr(md), ptr(vd), n, ptr(ss), max_norm, 1.0 / ranks, lr, b0, b1, eps,
         step, torch_form, stream())
    assert total > max_norm                                      # the clip is active
    assert rel_err(md.cpu().numpy(), m2) < 1e-5 and rel_err(vd.cpu().numpy(), v2) < 1e-5
    got = pd.cpu().numpy()
    assert np.abs(got - p2).max() <= np.spacing(np.abs(p).max())            # one fp32 ulp of the largest parameter
    small = np.abs(p) < 0.25                                                # where the update itself is resolved
    assert rel_err((got - p)[small], (p2 - p.astype(np.float64))[small]) < 2e-3


@pytest.mark.parametrize("B,M,K,P", [(2, 256, 32, 64 * 64), (1, 96, 510, 1000), (3, 70, 33, 36), (1, 512, 128, 4096)])
def test_linear_backward(B, M, K, P):
    """dW, db from the contraction kernel; dX = the forward GEMM on the transposed weights."""
    x, w, b, dy = rnd(B, K, P, seed=1), rnd(M, K, scale=K ** -0.5, seed=2), rnd(M, seed=3), rnd(B, M, P, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    y = torch.einsum("ok,bkp->bop", wt, xt) + bt[None, :, None]
    y.backward(torch.from_numpy(dy))
    xd, dyd = dev(x), dev(dy)
    dw = torch.full((M, K), 7.0, device="cuda")
    db = torch.full((M,), 7.0, device="cuda")
    call("jirf_train_linear_bwd_weight", ptr(dyd), ptr(xd), ptr(dw), ptr(db), B, M, K, P, 0, stream())
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL
    call("jirf_train_linear_bwd_weight", ptr(dyd), ptr(xd), ptr(dw), ptr(db), B, M, K, P, 1, stream())   # accumulate
    assert rel_err(dw.cpu().numpy(), 2 * wt.grad.numpy()) < REL
    dx = torch.empty(B, K, P, device="cuda")
    call("jirf_op_conv1x1", ptr(dyd), hptr(np.ascontiguousarray(w.T)), None, None, ptr(dx), B, M, K, P, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("B,Cn,P", [(2, 32, 1000), (1, 128, 300)])
def test_layernorm_backward(kind, B, Cn, P):
    x, w, b, dy = rnd(B, Cn, P, seed=1) * 2 + 0.5, 1 + 0.2 * rnd(Cn, seed=2), 0.1 * rnd(Cn, seed=3), rnd(B, Cn, P, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    y = oc.channel_layernorm(xt.reshape(B, Cn, P, 1), wt, bt, kind == 1)
    y.backward(torch.from_numpy(dy).reshape(B, Cn, P, 1))
    dx = torch.empty(B, Cn, P, device="cuda")
    dw, db = torch.zeros(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    xd, dyd, wd = dev(x), dev(dy), dev(w)
    call("jirf_train_layernorm_bwd", ptr(xd), 0, ptr(dyd), 0, ptr(wd), None, 0, ptr(dx), 0, ptr(dw), ptr(db), B, Cn, P,
         kind, 0, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    # forward with device weights, and the shortcut gradient added on the way out
    yd = torch.empty(B, Cn, P, device="cuda")
    bd = dev(b)
    call("jirf_train_layernorm", ptr(xd), 0, ptr(wd), ptr(bd) if kind == 1 else None, ptr(yd), 0, B, Cn, P, kind, stream())
    assert rel_err(yd.cpu().numpy(), y.detach().numpy().reshape(B, Cn, P)) < 1e-5
    call("jirf_train_layernorm_bwd", ptr(xd), 0, ptr(dyd), 0, ptr(wd), ptr(xd), 0, ptr(dx), 0, ptr(dw), ptr(db), B, Cn, P,
         kind, 1, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy() + x) < REL
    assert rel_err(dw.cpu().numpy(), 2 * wt.grad.numpy()) < REL
    if kind == 1:
        assert rel_err(db.cpu().numpy(), 2 * bt.grad.numpy()) < REL


def test_gelu_and_gate_backward():
    B, Cn, P = 2, 24, 500
    pre, dy = rnd(B, Cn, P, scale=1.5, seed=1), rnd(B, Cn, P, seed=2)
    pt = torch.from_numpy(pre).requires_grad_(True)
    oc.gelu(pt).backward(torch.from_numpy(dy))
    dx = torch.empty(B, Cn, P, device="cuda")
    pred_, dyd = dev(pre), dev(dy)
    call("jirf_train_gelu_bwd", ptr(pred_), 0, None, 0, ptr(dyd), 0, ptr(dx), 0, None, 0, B, Cn * P, 0, stream())
    assert rel_err(dx.cpu().numpy(), pt.grad.numpy()) < 1e-5
    yf = torch.empty(B, Cn, P, device="cuda")
    call("jirf_train_gelu", ptr(pred_), 0, None, 0, ptr(yf), 0, B, Cn * P, 0, stream())
    assert rel_err(yf.cpu().numpy(), oc.gelu(pt).detach().numpy()) < 1e-6
    # LeakyReLU(0.01) from its output
    lk = F.leaky_relu(pt.detach(), 0.01).numpy()
    lkd = dev(lk)
    call("jirf_train_gelu_bwd", ptr(lkd), 0, None, 0, ptr(dyd), 0, ptr(dx), 0, None, 0, B, Cn * P, 2, stream())
    assert rel_err(dx.cpu().numpy(), dy * np.where(pre > 0, 1.0, 0.01)) < 1e-6
    both = rnd(B, 2 * Cn, P, scale=1.5, seed=3)
    bt = torch.from_numpy(both).requires_grad_(True)
    a, b2 = bt.chunk(2, dim=1)
    (oc.gelu(a) * b2).backward(torch.from_numpy(dy))
    bd = dev(both)
    dx2 = torch.empty(B, 2 * Cn, P, device="cuda")
    half = 4 * Cn * P                                   # the two halves of one [B, 2C, P] tensor: batch stride 2 C P
    call("jirf_train_gelu_bwd", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(dyd), 0, ptr(dx2),
         2 * Cn * P, C.c_void_p(dx2.data_ptr() + half), 2 * Cn * P, B, Cn * P, 1, stream())
    assert rel_err(dx2.cpu().numpy(), bt.grad.numpy()) < 1e-5
    call("jirf_train_gelu", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(yf), 0, B, Cn * P, 1,
         stream())
    assert rel_err(yf.cpu().numpy(), (oc.gelu(a) * b2).detach().numpy()) < 1e-6
    # the gate on two raw halves (mode 3): y = gelu(a) * gelu(b), gradients with respect to both raw halves
    bt3 = torch.from_numpy(both).requires_grad_(True)
    a3, b3 = bt3.chunk(2, dim=1)
    (oc.gelu(a3) * oc.gelu(b3)).backward(torch.from_numpy(dy))
    call("jirf_train_gelu_bwd", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(dyd), 0, ptr(dx2),
         2 * Cn * P, C.c_void_p(dx2.data_ptr() + half), 2 * Cn * P, B, Cn * P, 3, stream())
    assert rel_err(dx2.cpu().numpy(), bt3.grad.numpy()) < 1e-5
    call("jirf_train_gelu", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(yf), 0, B, Cn * P, 3,
         stream())
    assert rel_err(yf.cpu().numpy(), (oc.gelu(a3) * oc.gelu(b3)).detach().numpy()) < 1e-6
    # y = a + b on slices, and the channel sum
    call("jirf_train_add", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(yf), 0, B, Cn * P, stream())
    assert rel_err(yf.cpu().numpy(), both[:, :Cn] + both[:, Cn:]) < 1e-6
    # y = a + s[image] * b (the drop-path residual; one fused multiply-add) and y = s[image] * b (bit-exact)
    sc = np.array([0.0, 1.0 / 0.9], np.float32)
    scd = dev(sc)
    call("jirf_train_scale_add", ptr(bd), 2 * Cn * P, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(scd), ptr(yf), 0, B,
         Cn * P, stream())
    assert rel_err(yf.cpu().numpy(), both[:, :Cn] + sc[:, None, None] * both[:, Cn:]) < 2e-7
    assert np.array_equal(yf.cpu().numpy()[0], both[0, :Cn])
    call("jirf_train_scale_add", None, 0, C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(scd), ptr(yf), 0, B, Cn * P, stream())
    assert np.array_equal(yf.cpu().numpy(), sc[:, None, None] * both[:, Cn:])
    cs = torch.zeros(Cn, device="cuda")
    call("jirf_train_channel_sum", C.c_void_p(bd.data_ptr() + half), 2 * Cn * P, ptr(cs), B, Cn, P, 0, stream())
    assert rel_err(cs.cpu().numpy(), both[:, Cn:].astype(np.float64).sum(axis=(0, 2))) < 1e-5


@pytest.mark.parametrize("B,Cn,H,W", [(2, 16, 24, 32), (1, 5, 9, 7)])
def test_depthwise_backward(B, Cn, H, W):
    x, w, b, dy = rnd(B, Cn, H, W, seed=1), rnd(Cn, 1, 3, 3, seed=2), rnd(Cn, seed=3), rnd(B, Cn, H, W, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    F.conv2d(xt, wt, bt, padding=1, groups=Cn).backward(torch.from_numpy(dy))
    dw, db = torch.zeros(Cn, 1, 3, 3, device="cuda"), torch.zeros(Cn, device="cuda")
    xd, dyd = dev(x), dev(dy)
    call("jirf_train_dwconv3x3_bwd_weight", ptr(xd), 0, ptr(dyd), 0, ptr(dw), ptr(db), B, Cn, H, W, 0, stream())
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL
    dx = torch.empty(B, Cn, H, W, device="cuda")
    wd = dev(w)
    call("jirf_train_dwconv3x3", ptr(dyd), 0, ptr(wd), None, ptr(dx), 0, B, Cn, H, W, 1, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL



W_FWD, W_BWD_DATA, W_DECONV, W_DOWN_BWD = 0, 1, 2, 4


def _conv(x, w, kind, bias, y, B, Cin, Cout, Hin, Win, ks=1, stride=1, pad=0, epi=0, r=None):
    call("jirf_train_conv", ptr(x), 0, ptr(w), kind, ptr(bias), ptr(r), 0, ptr(y), 0, B, Cin, Cout, Hin, Win, ks, stride, pad,
         epi, stream())


@pytest.mark.parametrize("B,Cin,Cout,H,ks,stride,pad", [(2, 8, 8, 16, 3, 1, 1), (1, 3, 32, 32, 3, 1, 1), (2, 64, 3, 16, 3, 1, 1),
                                                        (2, 32, 64, 16, 4, 2, 1), (1, 48, 20, 16, 1, 1, 0)])
def test_convolution_forward_and_both_gradients(B, Cin, Cout, H, ks, stride, pad):
    """nn.Conv2d on device weights (input / output projections ast_arch.py:558-565,594-600, FRFN's partial 3x3 :470,
    Downsample :630-636, Linear as ks = 1): forward, data gradient and weight / bias gradient against autograd."""
    Ho = (H + 2 * pad - ks) // stride + 1
    x, w, b = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, ks, ks, scale=(Cin * ks * ks) ** -0.5, seed=2), rnd(Cout, seed=3)
    dy = rnd(B, Cout, Ho, Ho, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    y = F.conv2d(xt, wt, bt, stride=stride, padding=pad)
    y.backward(torch.from_numpy(dy))
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    yd = torch.empty(B, Cout, Ho, Ho, device="cuda")
    _conv(xd, wd, W_FWD, bd, yd, B, Cin, Cout, H, H, ks, stride, pad)
    assert rel_err(yd.cpu().numpy(), y.detach().numpy()) < REL
    dw, db = torch.full((Cout, Cin, ks, ks), 3.0, device="cuda"), torch.full((Cout,), 3.0, device="cuda")
    call("jirf_train_conv_bwd_weight", ptr(dyd), 0, ptr(xd), 0, ptr(dw), ptr(db), B, Cout, Cin, Ho, Ho, H, H, ks, stride, pad,
         0, stream())
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL
    dx = torch.empty(B, Cin, H, H, device="cuda")
    if stride == 1:
        _conv(dyd, wd, W_BWD_DATA, None, dx, B, Cout, Cin, Ho, Ho, ks, 1, pad)
    else:
        _conv(dyd, wd, W_DOWN_BWD, None, dx, B, Cout, Cin, Ho, Ho)
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 64, 32, 8), (1, 512, 256, 16)])
def test_transposed_convolution_forward_and_both_gradients(B, Cin, Cout, H):
    """nn.ConvTranspose2d(k 2, stride 2) (Upsample ast_arch.py:653-660)."""
    x, w, b = rnd(B, Cin, H, H, seed=1), rnd(Cin, Cout, 2, 2, scale=Cin ** -0.5, seed=2), rnd(Cout, seed=3)
    dy = rnd(B, Cout, 2 * H, 2 * H, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    y = F.conv_transpose2d(xt, wt, bt, stride=2)
    y.backward(torch.from_numpy(dy))
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    yd = torch.empty(B, Cout, 2 * H, 2 * H, device="cuda")
    _conv(xd, wd, W_DECONV, bd, yd, B, Cin, Cout, H, H)
    assert rel_err(yd.cpu().numpy(), y.detach().numpy()) < REL
    dx = torch.empty(B, Cin, H, H, device="cuda")
    _conv(dyd, wd, W_FWD, None, dx, B, Cout, Cin, 2 * H, 2 * H, 2, 2, 0)
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL
    dw, db = torch.zeros(Cin, Cout, 2, 2, device="cuda"), torch.zeros(Cout, device="cuda")
    call("jirf_train_conv_bwd_weight", ptr(xd), 0, ptr(dyd), 0, ptr(dw), None, B, Cin, Cout, H, H, 2 * H, 2 * H, 2, 2, 0, 0,
         stream())
    call("jirf_train_channel_sum", ptr(dyd), 0, ptr(db), B, Cout, 4 * H * H, 0, stream())
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL


def test_pack_plan_equals_packing_inside_the_call():
    """One launch packs every (weight, kind) pair of a plan; jirf_train_conv on the packed buffers (w_kind |
    JIRF_W_PACKED) gives bit for bit what the same call gives when it packs its weight itself: all four kinds,
    3 x 3 / 1 x 1 / 4 x 4 stride 2 / transposed, odd channel counts, and again after the weights have changed."""
    from jirf_hip import lib
    L = lib()
    W_PACKED = 8
    B, H = 2, 16
    # (kind, Cin, Cout, ks, stride, pad, weight shape, Hin of the call)
    cases = [(W_FWD, 24, 40, 3, 1, 1, (40, 24, 3, 3), H), (W_FWD, 130, 70, 1, 1, 0, (70, 130), H),
             (W_FWD, 16, 32, 4, 2, 1, (32, 16, 4, 4), H), (W_BWD_DATA, 40, 24, 3, 1, 1, (40, 24, 3, 3), H),
             (W_BWD_DATA, 70, 130, 1, 1, 0, (70, 130), H), (W_DECONV, 64, 32, 1, 1, 0, (64, 32, 2, 2), 8),
             (W_DOWN_BWD, 32, 16, 1, 1, 0, (32, 16, 4, 4), 8)]
    ws = [dev(rnd(*c[6], scale=0.2, seed=10 + i)) for i, c in enumerate(cases)]
    bias = dev(rnd(32, seed=5))
    sizes = [int(L.jirf_train_packed_floats(c[0], c[1], c[2], c[3])) for c in cases]
    assert min(sizes) > 0 and L.jirf_train_packed_floats(3, 8, 8, 1) < 0
    bufs = [torch.full((n,), float("nan"), device="cuda") for n in sizes]
    n = len(cases)
    arr_p, arr_i = C.c_void_p * n, C.c_int * n
    plan = C.c_void_p()
    call("jirf_train_pack_plan_create", n, arr_p(*[w.data_ptr() for w in ws]),
         arr_p(*[bias.data_ptr() if c[0] == W_DECONV else None for c in cases]), arr_p(*[b.data_ptr() for b in bufs]),
         arr_i(*[c[0] for c in cases]), arr_i(*[c[1] for c in cases]), arr_i(*[c[2] for c in cases]),
         arr_i(*[c[3] for c in cases]), C.byref(plan))
    try:
        for trial in range(2):
            call("jirf_train_pack_plan_run", plan, stream())
            for (kind, Cin, Cout, ks, stride, pad, _, Hin), w, wt in zip(cases, ws, bufs):
                x = dev(rnd(B, Cin, Hin, Hin, seed=40 + trial))
                Ho = 2 * Hin if kind in (W_DECONV, W_DOWN_BWD) else (Hin + 2 * pad - ks) // stride + 1
                bd = bias if kind == W_DECONV else None
                y0 = torch.empty(B, Cout, Ho, Ho, device="cuda")
                y1 = torch.empty_like(y0)
                _conv(x, w, kind, bd, y0, B, Cin, Cout, Hin, Hin, ks, stride, pad)
                _conv(x, wt, kind | W_PACKED, bd, y1, B, Cin, Cout, Hin, Hin, ks, stride, pad)
                assert torch.isfinite(y1).all() and torch.equal(y0, y1), (kind, Cin, Cout, ks)
            for w in ws:                                  # a weight update: the next run repacks
                w.mul_(0.5).add_(0.01)
    finally:
        L.jirf_train_pack_plan_destroy(plan)


@pytest.mark.parametrize("coupled", [0, 1])
@pytest.mark.parametrize("B,heads,S,shift", [(2, 2, 16, 0), (2, 4, 16, 4), (1, 16, 8, 4), (3, 1, 24, 4)])
def test_window_attention_forward_and_backward(B, heads, S, shift, coupled):
    """WindowAttention_sparse (ast_arch.py:275-318) inside the shift / partition / reverse of TransformerBlock
    (:786-837), from the fused q | k | v tensor: output and the gradients of qkv, the bias table and the mixing
    weights against autograd over the oracle.  coupled = 0 is B separate calls, 1 the literal batched call."""
    Cn, P = 32 * heads, S * S
    qkv = rnd(B, 3 * Cn, P, scale=0.7, seed=1)
    table, w2 = rnd(225, heads, scale=0.3, seed=2), np.array([0.3, -0.2], np.float32)
    dout = rnd(B, Cn, P, seed=3)
    qt, tt, wt = (torch.from_numpy(a).requires_grad_(True) for a in (qkv, table, w2))

    def oracle(q3):                         # q3: [b, 3C, P] planar -> [b, C, P]
        b = q3.shape[0]
        tok = q3.reshape(b, 3 * Cn, S, S).permute(0, 2, 3, 1)                       # [b, S, S, 3C]
        if shift:
            tok = torch.roll(tok, shifts=(-shift, -shift), dims=(1, 2))
        wins = so.window_partition(tok, 8).reshape(-1, 64, 3 * Cn)
        B_ = wins.shape[0]
        q = wins[..., :Cn].reshape(B_, 64, heads, 32).permute(0, 2, 1, 3) * 32 ** -0.5
        k = wins[..., Cn:2 * Cn].reshape(B_, 64, heads, 32).permute(0, 2, 1, 3)
        v = wins[..., 2 * Cn:].reshape(B_, 64, heads, 32).permute(0, 2, 1, 3)
        mask = so.shift_attn_mask(S, S, 8, shift) if shift else None
        o = so.sparse_attention_core(q, k, v, tt, wt, 8, mask).reshape(B_, 8, 8, Cn)
        o = so.window_reverse(o, 8, S, S)
        if shift:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        return o.permute(0, 3, 1, 2).reshape(b, Cn, P)

    out = oracle(qt) if coupled else torch.cat([oracle(qt[i:i + 1]) for i in range(B)])
    out.backward(torch.from_numpy(dout))
    qd, td, wd, dod = dev(qkv), dev(table), dev(w2), dev(dout)
    od = torch.empty(B, Cn, P, device="cuda")
    call("jirf_train_window_attention", ptr(qd), 0, ptr(td), ptr(wd), ptr(od), 0, B, Cn, S, S, heads, shift, coupled, stream())
    assert rel_err(od.cpu().numpy(), out.detach().numpy()) < REL
    dq = torch.full((B, 3 * Cn, P), 7.0, device="cuda")
    dt, dw = torch.zeros(225, heads, device="cuda"), torch.zeros(2, device="cuda")
    call("jirf_train_window_attention_bwd", ptr(qd), 0, ptr(td), ptr(wd), ptr(dod), 0, ptr(dq), 0, ptr(dt), ptr(dw), B, Cn, S, S,
         heads, shift, coupled, 0, stream())
    for name, got, want in (("qkv", dq, qt.grad), ("table", dt, tt.grad), ("w", dw, wt.grad)):
        e = rel_err(got.cpu().numpy(), want.numpy())
        print(f"[window attention backward B={B} heads={heads} S={S} shift={shift} coupled={coupled}] {name}: {e:.2e}")
        assert e < REL, name
