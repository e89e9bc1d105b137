"""Ops shared by the oracle restatements (test infrastructure, parity unpinned).

All tensors are torch CPU tensors, fp32 unless stated.  Citations are relative
to the reference tree's ``basicsr/models/archs/``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, fields
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class JittorAssumptions:
    """The Jittor 1.3.10 behaviours the restatements rely on and cannot check offline (Jittor is neither
    vendored by the reference nor installable here), each under a name.  `ASSUMPTIONS` holds the active
    set; tests/conftest.py prints it once per session.  Fields marked [fixed] have one implemented reading
    (the alternative is stated for the record); the others switch the helpers of this module, and
    HintSwitches / FproSwitches hold the per-network ones (floor division, regroup precision, bicubic
    kernel, ifft scaling)."""
    # 1 [fixed] Var.transpose(a, b) with two ints swaps the two axes (ast_arch.py:179,280; hint_arch.py:29,364):
    #   bmm_4d's shape test only passes on that reading
    transpose_two_ints: str = "swap"
    # 2 jt.normalize(x, dim=-1) = x / sqrt(max(sum x^2, eps)) (hint_arch.py:361-362; astv2_arch.py:174-175)
    normalize_eps: float = 1e-30
    # 4 nn.GELU / nn.gelu: "erf" (0.5 x (1 + erf(x / sqrt 2))) or "tanh"
    gelu: str = "erf"
    # 5, 9 LayerNorm: eps inside the sqrt, "population" or "unbiased" variance
    layernorm_eps: float = 1e-5
    variance: str = "population"
    # 6 nn.LeakyReLU() negative slope (ast_arch.py:608,619)
    leaky_relu_slope: float = 0.01
    # 7 [fixed] nn.interpolate(bilinear, align_corners=False): half-pixel centres, edge clamp (hint_arch.py:283-288)
    bilinear: str = "half_pixel"
    # 8 [fixed] jt.sort(descending=True) returns (values, indices); ties in index order (hint_arch.py:249)
    sort_ties: str = "stable"
    # 10 nn.PixelShuffle(2) channel order: "torch" (in[n, 4c + 2i + j]) or "grouped" (in[n, (2i + j) C + c])
    pixel_shuffle_order: str = "torch"
    # 11, 12 [fixed] Conv / Conv1d weights [out, in/groups, k..], ConvTranspose [in, out, kh, kw], Linear [out, in]
    weight_layout: str = "torch"
    # 14 [fixed] nn.pad(reflect), jt.roll, chunk / split / concat, softmax (max-subtracted) behave like torch
    tensor_ops: str = "torch"

    def describe(self):
        return [f"{f.name}={getattr(self, f.name)!r}" for f in fields(self)]


ASSUMPTIONS = JittorAssumptions()


def as_params(state: dict) -> Params:
    """numpy / torch state dict -> dict of contiguous CPU torch tensors.  Tensors that require a gradient are
    passed through as they are (the training tests differentiate the oracle with torch autograd)."""
    out = {}
    for k, v in state.items():
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        out[k] = t if t.requires_grad else t.detach().cpu().contiguous()
    return out


def gelu(x: torch.Tensor) -> torch.Tensor:
    """GELU in the form ASSUMPTIONS.gelu names; the default is the exact (erf) one, 0.5 x (1 + erf(x / sqrt 2))."""
    return F.gelu(x, approximate="none" if ASSUMPTIONS.gelu == "erf" else "tanh")


def leaky_relu(x: torch.Tensor) -> torch.Tensor:
    return F.leaky_relu(x, ASSUMPTIONS.leaky_relu_slope)


def _variance(x: torch.Tensor, mu: torch.Tensor, dim: int) -> torch.Tensor:
    n = x.shape[dim]
    ss = ((x - mu) ** 2).sum(dim=dim, keepdim=True)
    return ss / (n if ASSUMPTIONS.variance == "population" else n - 1)


def token_layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm(C) over the last axis (ast_arch.py:694,718,743)."""
    if ASSUMPTIONS.variance == "population":
        return F.layer_norm(x, (x.shape[-1],), weight, bias, ASSUMPTIONS.layernorm_eps)
    mu = x.mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(_variance(x, mu, -1) + ASSUMPTIONS.layernorm_eps) * weight + bias


def channel_layernorm(x: torch.Tensor, weight: torch.Tensor, bias, with_bias: bool,
                      eps: float = None) -> torch.Tensor:
    """Per-pixel LayerNorm over the channel axis of an NCHW tensor.

    hint_arch.py:98-145 / astv2_arch.py:7-74: population variance, eps inside
    the sqrt.  The bias-free form divides the UN-centred x by sqrt(var+eps).
    """
    eps = ASSUMPTIONS.layernorm_eps if eps is None else eps
    mu = x.mean(dim=1, keepdim=True)
    var = _variance(x, mu, 1)
    w = weight.view(1, -1, 1, 1)
    if with_bias:
        return (x - mu) / torch.sqrt(var + eps) * w + bias.view(1, -1, 1, 1)
    return x / torch.sqrt(var + eps) * w


def pixel_unshuffle2(x: torch.Tensor) -> torch.Tensor:
    """out[n, 4c+2i+j, y, x] = in[n, c, 2y+i, 2x+j]  (hint_arch.py:48-56, astv2_arch.py:245-268)."""
    b, c, h, w = x.shape
    return (x.reshape(b, c, h // 2, 2, w // 2, 2)
             .permute(0, 1, 3, 5, 2, 4)
             .reshape(b, c * 4, h // 2, w // 2))


def pixel_shuffle2(x: torch.Tensor) -> torch.Tensor:
    """out[n, c, 2y+i, 2x+j] = in[n, 4c+2i+j, y, x]  (nn.PixelShuffle(2))."""
    b, c4, h, w = x.shape
    c = c4 // 4
    if ASSUMPTIONS.pixel_shuffle_order == "grouped":      # in[n, (2i + j) C + c, y, x]
        return (x.reshape(b, 2, 2, c, h, w)
                 .permute(0, 3, 4, 1, 5, 2)
                 .reshape(b, c, h * 2, w * 2))
    return (x.reshape(b, c, 2, 2, h, w)
             .permute(0, 1, 4, 2, 5, 3)
             .reshape(b, c, h * 2, w * 2))


def conv1x1(x: torch.Tensor, w: torch.Tensor, b=None) -> torch.Tensor:
    return F.conv2d(x, w, b)


def conv3x3(x: torch.Tensor, w: torch.Tensor, b=None, groups: int = 1) -> torch.Tensor:
    return F.conv2d(x, w, b, stride=1, padding=1, groups=groups)


def adaptive_pool_bounds(n_in: int, n_out: int):
    """[start, end) of every output cell: floor(i*n_in/n_out), ceil((i+1)*n_in/n_out).

    hint_arch.py:79-85.  Integer arithmetic gives the same values as the
    reference's float expression (the quotient is either an exact integer or at
    least 1/n_out away from one).
    """
    starts = [(i * n_in) // n_out for i in range(n_out)]
    ends = [-((-(i + 1) * n_in) // n_out) for i in range(n_out)]
    return starts, ends


def adaptive_avg_pool_matrix(n_in: int, n_out: int) -> torch.Tensor:
    """[n_out, n_in] float64 matrix with 1 inside each output cell's range (un-normalised)."""
    s, e = adaptive_pool_bounds(n_in, n_out)
    m = torch.zeros(n_out, n_in, dtype=torch.float64)
    for i in range(n_out):
        m[i, s[i]:e[i]] = 1.0
    return m


def adaptive_avg_pool_sq(x: torch.Tensor, n_out: int) -> torch.Tensor:
    """Adaptive average pool of the last two axes to n_out x n_out (hint_arch.py:59-95).

    Region sums are taken in float64 and divided by the region's element count,
    then rounded to fp32 once: within 1 ulp of any fp32 evaluation order.
    """
    hin, win = x.shape[-2:]
    ph = adaptive_avg_pool_matrix(hin, n_out)
    pw = adaptive_avg_pool_matrix(win, n_out)
    cnt = ph.sum(1)[:, None] * pw.sum(1)[None, :]
    y = ph @ x.double() @ pw.t()
    return (y / cnt).float()


def psnr(a: np.ndarray, b: np.ndarray, peak: float = 1.0) -> float:
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)
