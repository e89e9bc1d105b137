"""Backward / optimiser kernels of the training step against torch autograd over the oracle's operators
(torch-CPU fp32; the reference's Jittor autograd cannot run here, so this is "parity unpinned" like the
forward): gradients within 1e-3 relative of autograd's, the loss and the Adam update against a NumPy
restatement of Jittor's optimizer (jt.optim.Adam.step, Optimizer.clip_grad_norm).  Ends with the backward
of one whole FRFN token MLP (ast_arch.py:476-513) composed from the kernels, against autograd over
oracle.ast_oracle.frfn: the operators that hold most of AST's parameters."""
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
    call("jirf_train_smooth_l1", ptr(pd_), ptr(td_), ptr(g), n, 0.7, C.byref(out), stream())
    assert abs(out.value - float(loss)) < 1e-6 * abs(float(loss))
    assert rel_err(g.cpu().numpy(), pt.grad.numpy()) < 1e-6
    assert (np.abs(pred - tgt) > 1).any() and (np.abs(pred - tgt) < 1).any()      # both branches


@pytest.mark.parametrize("torch_form", [0, 1])
def test_clip_and_adam_step(torch_form):
    n = 100003
    p, g = rnd(n, seed=3), rnd(n, scale=1e-3, seed=4)
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
    pd, gd, md, vd = dev(p), dev(g), dev(m), dev(v)
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    hs = C.c_double()
    call("jirf_train_sumsq", ptr(gd), n, ptr(ss), C.byref(hs), stream())
    assert abs(hs.value - total ** 2) < 1e-9 * total ** 2
    call("jirf_train_adam_step", ptr(pd), ptr(gd), ptr(md), ptr(vd), n, ptr(ss), max_norm, lr, b0, b1, eps, step,
         torch_form, stream())
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
    call("jirf_train_layernorm_bwd", ptr(xd), ptr(dyd), ptr(wd), ptr(dx), ptr(dw), ptr(db), B, Cn, P, kind, 0, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    if kind == 1:
        assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL


def test_gelu_and_gate_backward():
    B, Cn, P = 2, 24, 500
    pre, dy = rnd(B, Cn, P, scale=1.5, seed=1), rnd(B, Cn, P, seed=2)
    pt = torch.from_numpy(pre).requires_grad_(True)
    oc.gelu(pt).backward(torch.from_numpy(dy))
    dx = torch.empty(B, Cn, P, device="cuda")
    pred_, dyd = dev(pre), dev(dy)
    call("jirf_train_gelu_bwd", ptr(pred_), None, ptr(dyd), ptr(dx), B, Cn, P, 0, stream())
    assert rel_err(dx.cpu().numpy(), pt.grad.numpy()) < 1e-5
    both = rnd(B, 2 * Cn, P, scale=1.5, seed=3)
    bt = torch.from_numpy(both).requires_grad_(True)
    a, b2 = bt.chunk(2, dim=1)
    (oc.gelu(a) * b2).backward(torch.from_numpy(dy))
    bd = dev(both)
    dx2 = torch.empty(B, 2 * Cn, P, device="cuda")
    call("jirf_train_gelu_bwd", ptr(bd), C.c_void_p(bd.data_ptr() + 4 * Cn * P), ptr(dyd), ptr(dx2), B, Cn, P, 1, stream())
    assert rel_err(dx2.cpu().numpy(), bt.grad.numpy()) < 1e-5


@pytest.mark.parametrize("B,Cn,H,W", [(2, 16, 24, 32), (1, 5, 9, 7)])
def test_depthwise_backward(B, Cn, H, W):
    x, w, b, dy = rnd(B, Cn, H, W, seed=1), rnd(Cn, 1, 3, 3, seed=2), rnd(Cn, seed=3), rnd(B, Cn, H, W, seed=4)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    F.conv2d(xt, wt, bt, padding=1, groups=Cn).backward(torch.from_numpy(dy))
    dw, db = torch.zeros(Cn, 1, 3, 3, device="cuda"), torch.zeros(Cn, device="cuda")
    xd, dyd = dev(x), dev(dy)
    call("jirf_train_dwconv3x3_bwd_weight", ptr(xd), ptr(dyd), ptr(dw), ptr(db), B, Cn, H, W, 0, stream())
    assert rel_err(dw.cpu().numpy(), wt.grad.numpy()) < REL
    assert rel_err(db.cpu().numpy(), bt.grad.numpy()) < REL
    dx = torch.empty(B, Cn, H, W, device="cuda")
    wfd = dev(np.ascontiguousarray(w[:, :, ::-1, ::-1]))
    call("jirf_op_dwconv3x3", ptr(dyd), ptr(wfd), None, ptr(dx), B, Cn, H, W, 0, stream())
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < REL


def test_frfn_token_mlp_backward_composed_from_the_kernels():
    """x -> FRFN(x) (ast_arch.py:476-513) with every gradient (input, linear1, dwconv, linear2, partial_conv3
    excluded: its dense 3x3 weight gradient is not built yet) from the device kernels, against autograd over
    oracle.ast_oracle.frfn.  Planar layout: tokens [B, L, C] are [B, C, L] here."""
    B, Cn, S = 2, 32, 16
    h, P = 4 * Cn, S * S
    rng = np.random.default_rng(5)
    p = {"m.linear1.0.weight": rnd(2 * h, Cn, scale=Cn ** -0.5, seed=1), "m.linear1.0.bias": 0.1 * rnd(2 * h, seed=2),
         "m.dwconv.0.weight": rnd(h, 1, 3, 3, scale=0.3, seed=3), "m.dwconv.0.bias": 0.1 * rnd(h, seed=4),
         "m.linear2.0.weight": rnd(Cn, h, scale=h ** -0.5, seed=5), "m.linear2.0.bias": 0.1 * rnd(Cn, seed=6),
         "m.partial_conv3.weight": rnd(Cn // 4, Cn // 4, 3, 3, scale=0.2, seed=7)}
    x = rng.standard_normal((B, P, Cn)).astype(np.float32)
    dy = rng.standard_normal((B, P, Cn)).astype(np.float32)
    pt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in p.items()}
    xt = torch.from_numpy(x).requires_grad_(True)
    so.frfn(pt, "m", xt).backward(torch.from_numpy(dy))

    # ---- forward on the CPU up to the stashes a training forward keeps (the forward kernels are tested elsewhere)
    with torch.no_grad():
        t = xt.reshape(B, S, S, Cn).permute(0, 3, 1, 2)
        cat = torch.cat([F.conv2d(t[:, :Cn // 4], pt["m.partial_conv3.weight"], None, padding=1), t[:, Cn // 4:]], 1)
        cat_p = cat.reshape(B, Cn, P)                                                  # input of linear1, planar
        pre1 = torch.einsum("ok,bkp->bop", pt["m.linear1.0.weight"], cat_p) + pt["m.linear1.0.bias"][None, :, None]
        act1 = oc.gelu(pre1)                                                           # [B, 2h, P]: x1 | x2
        pre_dw = F.conv2d(act1[:, :h].reshape(B, h, S, S), pt["m.dwconv.0.weight"], pt["m.dwconv.0.bias"], padding=1,
                          groups=h).reshape(B, h, P)
        gated = oc.gelu(pre_dw) * act1[:, h:]
    dyp = np.ascontiguousarray(dy.transpose(0, 2, 1))                                   # planar [B, C, P]
    dyd = dev(dyp)
    # linear2
    dw2, db2 = torch.zeros(Cn, h, device="cuda"), torch.zeros(Cn, device="cuda")
    gated_d = dev(gated.numpy())
    call("jirf_train_linear_bwd_weight", ptr(dyd), ptr(gated_d), ptr(dw2), ptr(db2), B, Cn, h, P, 0, stream())
    dgated = torch.empty(B, h, P, device="cuda")
    call("jirf_op_conv1x1", ptr(dyd), hptr(np.ascontiguousarray(p["m.linear2.0.weight"].T)), None, None, ptr(dgated),
         B, Cn, h, P, stream())
    # gate: (d pre_dw, d x2) from (pre_dw | x2)
    gate_in = dev(np.concatenate([pre_dw.numpy(), act1[:, h:].numpy()], axis=1))
    dgate = torch.empty(B, 2 * h, P, device="cuda")
    call("jirf_train_gelu_bwd", ptr(gate_in), C.c_void_p(gate_in.data_ptr() + 4 * h * P), ptr(dgated), ptr(dgate), B, h, P,
         1, stream())
    # depthwise: weight / bias gradients and d x1
    dwdw, dbdw = torch.zeros(h, 1, 3, 3, device="cuda"), torch.zeros(h, device="cuda")
    x1d = dev(np.ascontiguousarray(act1[:, :h].numpy()))
    dpre_dw = dgate[:, :h].contiguous()
    call("jirf_train_dwconv3x3_bwd_weight", ptr(x1d), ptr(dpre_dw), ptr(dwdw), ptr(dbdw), B, h, S, S, 0, stream())
    dact1 = torch.empty(B, 2 * h, P, device="cuda")
    dx1 = torch.empty(B, h, P, device="cuda")
    wfd = dev(np.ascontiguousarray(p["m.dwconv.0.weight"][:, :, ::-1, ::-1]))
    call("jirf_op_dwconv3x3", ptr(dpre_dw), ptr(wfd), None, ptr(dx1), B, h, S, S, 0, stream())
    dact1[:, :h] = dx1
    dact1[:, h:] = dgate[:, h:]
    # GELU after linear1, then linear1
    dpre1 = torch.empty(B, 2 * h, P, device="cuda")
    pre1_d = dev(pre1.numpy())
    call("jirf_train_gelu_bwd", ptr(pre1_d), None, ptr(dact1), ptr(dpre1), B, 2 * h, P, 0, stream())
    dw1, db1 = torch.zeros(2 * h, Cn, device="cuda"), torch.zeros(2 * h, device="cuda")
    cat_d = dev(cat_p.numpy())
    call("jirf_train_linear_bwd_weight", ptr(dpre1), ptr(cat_d), ptr(dw1), ptr(db1), B, 2 * h, Cn, P, 0, stream())
    dcat = torch.empty(B, Cn, P, device="cuda")
    call("jirf_op_conv1x1", ptr(dpre1), hptr(np.ascontiguousarray(p["m.linear1.0.weight"].T)), None, None, ptr(dcat),
         B, 2 * h, Cn, P, stream())
    # partial conv: the first C/4 channels go back through the dense 3x3 (transposed, flipped weights), the rest pass
    wpc = p["m.partial_conv3.weight"]
    wpc_t = np.ascontiguousarray(wpc.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
    dfirst = torch.empty(B, Cn // 4, P, device="cuda")
    dcat_first = dcat[:, :Cn // 4].contiguous()
    call("jirf_op_conv3x3", ptr(dcat_first), hptr(wpc_t), None, None, ptr(dfirst), B, Cn // 4, Cn // 4, S, S, 0, stream())
    dx = torch.cat([dfirst, dcat[:, Cn // 4:]], dim=1).cpu().numpy().transpose(0, 2, 1)

    checks = {"linear2.weight": (dw2, pt["m.linear2.0.weight"].grad), "linear2.bias": (db2, pt["m.linear2.0.bias"].grad),
              "dwconv.weight": (dwdw, pt["m.dwconv.0.weight"].grad), "dwconv.bias": (dbdw, pt["m.dwconv.0.bias"].grad),
              "linear1.weight": (dw1, pt["m.linear1.0.weight"].grad), "linear1.bias": (db1, pt["m.linear1.0.bias"].grad)}
    for name, (got, want) in checks.items():
        e = rel_err(got.cpu().numpy(), want.numpy())
        print(f"[frfn backward] {name}: relative error {e:.2e}")
        assert e < REL, name
    e = rel_err(dx, xt.grad.numpy())
    print(f"[frfn backward] input: relative error {e:.2e}")
    assert e < REL
