"""The gated feed-forward branch (LayerNorm -> project_in -> depthwise 3x3 -> gelu gate -> project_out -> residual;
reference FeedForward.execute hint_arch.py:180-186 inside TransformerBlock.execute :429-433) through the C ABI,
in its one-kernel form (the hidden tensor stays on the compute unit) and as the three launches, both against the
oracle's operators on the same weights: fp32, max-abs <= 1e-3 (observed ~1e-6)."""
import numpy as np
import pytest
import torch

from oracle import common as oc
from oracle import hint_oracle as ho
from tests.gpu_util import call, dev, hptr, max_abs, ptr, stream

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _case(B, C, h, H, W, bias, seed):
    rng = np.random.default_rng(seed)
    f = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    p = {"n.weight": (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), "n.bias": f(C, scale=0.2),
         "ffn.project_in.weight": f(2 * h, C, 1, 1, scale=C ** -0.5), "ffn.dwconv.weight": f(2 * h, 1, 3, 3, scale=0.4),
         "ffn.project_out.weight": f(C, h, 1, 1, scale=h ** -0.5)}
    if bias:
        p.update({"ffn.project_in.bias": f(2 * h, scale=0.3), "ffn.dwconv.bias": f(2 * h, scale=0.3),
                  "ffn.project_out.bias": f(C, scale=0.3)})
    x = (f(B, C, H, W) * 1.5 + 0.3).astype(np.float32)
    return p, x


def _oracle(p, x, ln_kind):
    t = {k: torch.from_numpy(v) for k, v in p.items()}
    xt = torch.from_numpy(x)
    n = oc.channel_layernorm(xt, t["n.weight"], t["n.bias"], ln_kind == 1)
    return (xt + ho.gdfn(t, "ffn", n)).numpy()


def _run(p, x, h, ln_kind, fused):
    B, C, H, W = x.shape
    xd = dev(x)
    y = torch.full((B, C, H, W), float("nan"), device="cuda")
    g = lambda k: hptr(np.ascontiguousarray(p[k])) if k in p else None
    keep = [np.ascontiguousarray(v) for v in p.values()]          # (the pointers above must outlive the call)
    call("jirf_op_gdfn", ptr(xd), ptr(y), B, C, h, H, W, ln_kind, hptr(p["n.weight"]), hptr(p["n.bias"]),
         hptr(p["ffn.project_in.weight"]), g("ffn.project_in.bias"), hptr(p["ffn.dwconv.weight"]), g("ffn.dwconv.bias"),
         hptr(p["ffn.project_out.weight"]), g("ffn.project_out.bias"), fused, stream())
    del keep
    return y.cpu().numpy()


@pytest.mark.parametrize("B,h,H,W,ln_kind,bias", [
    (1, 255, 40, 64, 1, False),      # HINT's width at 96 channels; three strips, the last one partly outside
    (2, 255, 33, 95, 1, False),      # odd plane: rows and columns that no tile size divides
    (1, 255, 64, 32, 2, False),      # one strip wide (both edges in one strip), BiasFree LayerNorm
    (1, 200, 24, 61, 1, True),       # a hidden width that fills no whole chunk, every bias present
    (1, 256, 17, 128, 2, True),      # the widest hidden tensor it takes
    (3, 255, 128, 128, 1, False),    # HINT level 2 of a 256 x 256 crop
])
def test_gdfn_one_kernel_form(B, h, H, W, ln_kind, bias):
    p, x = _case(B, 96, h, H, W, bias, seed=B + h + H + W)
    p = {k: np.ascontiguousarray(v) for k, v in p.items()}
    want = _oracle(p, x, ln_kind)
    got = _run(p, x, h, ln_kind, fused=1)
    three = _run(p, x, h, ln_kind, fused=0)
    e1, e3, e13 = max_abs(got, want), max_abs(three, want), max_abs(got, three)
    print(f"[gdfn B{B} h{h} {H}x{W} ln{ln_kind} bias={bias}] one kernel vs oracle {e1:.2e}, three launches vs oracle {e3:.2e}, "
          f"one vs three {e13:.2e}; |y| max {np.abs(want).max():.2f}")
    assert np.isfinite(got).all()
    assert e1 < TOL and e3 < TOL


def test_gdfn_one_kernel_form_rejects_what_it_is_not_built_for():
    import jirf_hip
    L = jirf_hip.lib()
    z = torch.zeros(8, device="cuda")
    a = np.zeros(8, np.float32)
    rc = L.jirf_op_gdfn(ptr(z), ptr(z), 1, 96, 255, 8, 8, 1, hptr(a), hptr(a), hptr(a), None, hptr(a), None, hptr(a), None, 1,
                        stream())
    assert rc == -1                                                  # x and y alias
    y = torch.zeros(8, device="cuda")
    rc = L.jirf_op_gdfn(ptr(z), ptr(y), 1, 48, 127, 64, 64, 1, hptr(a), hptr(a), hptr(a), None, hptr(a), None, hptr(a), None, 1,
                        stream())
    assert rc == -1 and "one-kernel form" in jirf_hip.last_error()   # 48 channels: the three-launch form serves them
