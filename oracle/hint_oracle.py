"""Oracle for the HINT forward (test infrastructure, parity unpinned).

Restates ``basicsr/models/archs/hint_arch.py`` of the reference in torch-CPU
fp32.  Line numbers in the docstrings refer to that file.

Semantic switches (``HintSwitches``) name the Jittor behaviours that cannot be
checked offline; the defaults are the ones this build assumes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .common import (Params, adaptive_avg_pool_sq, as_params, channel_layernorm,
                     conv1x1, conv3x3, gelu, pixel_shuffle2, pixel_unshuffle2)

GROUP_RATIOS = (1, 2, 2, 3)          # Attention.group, :317
CACHE_MOMENTUM = 0.9                 # update_factor, :391

Tap = Optional[Callable[[str, torch.Tensor], None]]


@dataclass
class HintSwitches:
    # precision of the channel-correlation statistics that drive the argsort:
    # "fp64" = rounding-free value of the reference's formula (default, what the
    # HIP path computes); "fp32" = literal fp32 evaluation like the reference.
    regroup_precision: str = "fp64"
    # `Var // int` on floats: "floor" (torch semantics), "trunc" or "true".
    floordiv: str = "floor"
    # eps of jt.normalize: x / sqrt(max(sum x^2, eps))
    normalize_eps: float = 1e-30


def group_sizes(c: int) -> List[int]:
    """:259-260 — int(ratio / total * C) for the ratios (1, 2, 2, 3)."""
    tot = sum(GROUP_RATIOS)
    return [int(r / tot * c) for r in GROUP_RATIOS]


def default_config() -> dict:
    """network_g block of Denoising/Options/GaussianColorDenoising_HINT.yml:51-62."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                num_refinement_blocks=4, heads=[8, 8, 8, 8], ffn_expansion_factor=2.66,
                bias=False, ln_type="WithBias", dual_pixel_task=False)


def stage_plan(cfg: dict):
    """(name, n_blocks, dim multiplier, is_att) in execution order (:514-637).

    The latent stage uses num_blocks[1] (not [3]) blocks and decoder_level1 /
    refinement run at 2*dim.
    """
    nb = cfg["num_blocks"]
    return [
        ("encoder_level1", nb[0], 1, False),
        ("encoder_level2", nb[1], 2, False),
        ("encoder_level3", nb[2], 4, False),
        ("latent", nb[1], 8, True),
        ("decoder_level3", nb[2], 4, True),
        ("decoder_level2", nb[1], 2, True),
        ("decoder_level1", nb[0], 2, True),
        ("refinement", cfg["num_refinement_blocks"], 2, True),
    ]


# --------------------------------------------------------------------------
# pieces of the attention block
# --------------------------------------------------------------------------

def corr_mean_similarity(feat: torch.Tensor, precision: str) -> torch.Tensor:
    """Row means of the channel correlation matrix of feat [C, L]  (:24-32, :246-248)."""
    x = feat.double() if precision == "fp64" else feat
    xm = x - x.mean(dim=1, keepdim=True)
    cov = (xm @ xm.t()) / (x.shape[1] - 1)
    d = torch.sqrt(torch.diagonal(cov))
    corr = cov / (d[None, :] * d[:, None])
    return corr.mean(dim=1)


def regroup_permutation(q: torch.Tensor, precision: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Channel order of ReGroup (:244-249): descending mean correlation of the
    BATCH-MEAN of q [B, C, L]; ties keep ascending channel index."""
    sim = corr_mean_similarity(q.mean(dim=0), precision)
    order = torch.sort(sim, descending=True, stable=True).indices
    return order, sim


def l2_normalize_rows(x: torch.Tensor, eps: float) -> torch.Tensor:
    """jt.normalize(x, dim=-1): x / sqrt(max(sum x^2, eps))  (:361-362)."""
    n2 = (x * x).sum(dim=-1, keepdim=True)
    return x / torch.sqrt(torch.clamp(n2, min=eps))


def conv1d_k1(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.Conv1d(kernel_size=1) on [B, Cin, N]: z[b,o,n] = sum_i w[o,i,0] x[b,i,n] + b[o]."""
    return torch.einsum("oi,bin->bon", w[:, :, 0], x) + b.view(1, -1, 1)


def inter_modulation(p: Params, pre: str, attn: torch.Tensor, cache: torch.Tensor) -> torch.Tensor:
    """Inter_CacheModulation.execute (:205-210); attn [B,c,c], cache [1,C,C]."""
    c = attn.shape[-1]
    pooled = adaptive_avg_pool_sq(cache, c)                      # self.align, [1,c,c]
    z = conv1d_k1(pooled + attn, p[pre + ".conv_width.weight"], p[pre + ".conv_width.bias"])
    scale, shift = z[:, :c], z[:, c:]
    y = attn * scale + shift
    return y * gelu(conv1d_k1(y, p[pre + ".gating_conv.weight"], p[pre + ".gating_conv.bias"]))


def intra_modulation(p: Params, pre: str, out: torch.Tensor, tcache: torch.Tensor) -> torch.Tensor:
    """Intra_CacheModulation.execute (:225-228); out, tcache [B,C,L]."""
    s = tcache + out
    g = gelu(conv1d_k1(s, p[pre + ".gating_conv.weight"], p[pre + ".gating_conv.bias"])) * s
    d = conv1d_k1(g, p[pre + ".down.weight"], p[pre + ".down.bias"])
    return conv1d_k1(d, p[pre + ".up.weight"], p[pre + ".up.bias"])


def float_floordiv(x: torch.Tensor, d: int, mode: str) -> torch.Tensor:
    if mode == "floor":
        return torch.floor(x / d)
    if mode == "trunc":
        return torch.trunc(x / d)
    return x / d


def layer_cache(att: List[torch.Tensor], c: int, sw: HintSwitches) -> torch.Tensor:
    """calculate_layer_cache (:272-296): sum_g floor(bilinear(mean_B att_g) * c_g / C) -> [1,C,C]."""
    total = None
    for a in att:
        m = a.mean(dim=0, keepdim=True)                              # [1,c,c]
        up = F.interpolate(m.unsqueeze(1), size=(c, c), mode="bilinear", align_corners=False)
        term = float_floordiv(up * a.shape[-1], c, sw.floordiv)
        total = term if total is None else total + term
    return total.squeeze(1)


def hmha_core(p: Params, pre: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
              cache: Optional[torch.Tensor], sw: HintSwitches, tap: Tap = None,
              forced_order: Optional[torch.Tensor] = None):
    """The part of Attention.execute between the convolutions and the intra-modulator
    (:354-399).  q, k, v are [B,C,L].  Returns (out [B,C,L] in sorted channel order,
    tcache [B,C,L], att list, new cache [1,C,C], channel order)."""
    c = q.shape[1]
    order, sim = regroup_permutation(q, sw.regroup_precision)
    if tap:
        tap(pre + ".perm", order.clone())
        tap(pre + ".mean_similarity", sim.clone())
    if forced_order is not None:
        order = forced_order
    qs, ks, vs = q[:, order], k[:, order], v[:, order]

    sizes = group_sizes(c)
    temperature = p[pre + ".temperature"]
    att, tnorm = [], []
    s0 = 0
    for g, cg in enumerate(sizes):
        qn = l2_normalize_rows(qs[:, s0:s0 + cg], sw.normalize_eps)
        kn = l2_normalize_rows(ks[:, s0:s0 + cg], sw.normalize_eps)
        a = torch.matmul(qn, kn.transpose(-2, -1)) * temperature[g]
        att.append(torch.softmax(a, dim=-1))
        tnorm.append(qn + kn)
        s0 += cg
    tcache = torch.cat(tnorm, dim=1)

    if cache is not None and cache.shape[-1] != c:
        cache = adaptive_avg_pool_sq(cache, c)                       # :377-379

    outs = []
    s0 = 0
    for g, cg in enumerate(sizes):
        if cache is not None:
            att[g] = inter_modulation(p, f"{pre}.inter_modulator{g + 1}", att[g], cache) + att[g]
        outs.append(torch.matmul(att[g], vs[:, s0:s0 + cg]))
        s0 += cg
    if tap:
        for g in range(4):
            tap(f"{pre}.att{g}", att[g].clone())

    upd = layer_cache(att, c, sw)
    if cache is not None:
        cache = cache * CACHE_MOMENTUM + upd * (1 - CACHE_MOMENTUM)
    else:
        cache = upd * CACHE_MOMENTUM
    return torch.cat(outs, dim=1), tcache, att, cache, order          # out stays in sorted order


def hmha(p: Params, pre: str, x: torch.Tensor, cache: Optional[torch.Tensor],
         sw: HintSwitches, tap: Tap = None, forced_order=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Attention.execute (:345-405). x is the normalised input [B,C,H,W]."""
    b, c, h, w = x.shape
    qkv = conv3x3(conv1x1(x, p[pre + ".qkv.weight"], p.get(pre + ".qkv.bias")),
                  p[pre + ".qkv_dwconv.weight"], p.get(pre + ".qkv_dwconv.bias"), groups=3 * c)
    q, k, v = [t.reshape(b, c, h * w) for t in qkv.chunk(3, dim=1)]
    out, tcache, _, cache, _ = hmha_core(p, pre, q, k, v, cache, sw, tap, forced_order)
    out = intra_modulation(p, pre + ".intra_modulator", out, tcache) + out
    out = conv1x1(out.reshape(b, c, h, w), p[pre + ".project_out.weight"],
                  p.get(pre + ".project_out.bias"))
    return out, cache


def gdfn(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward.execute (:180-186)."""
    t = conv1x1(x, p[pre + ".project_in.weight"], p.get(pre + ".project_in.bias"))
    t = conv3x3(t, p[pre + ".dwconv.weight"], p.get(pre + ".dwconv.bias"), groups=t.shape[1])
    x1, x2 = t.chunk(2, dim=1)
    return conv1x1(gelu(x1) * x2, p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def transformer_block(p: Params, pre: str, x: torch.Tensor, cache, is_att: bool,
                      with_bias_ln: bool, sw: HintSwitches, tap: Tap = None, forced_order=None):
    """TransformerBlock.execute (:426-434)."""
    if is_att:
        n1 = channel_layernorm(x, p[pre + ".norm1.body.weight"], p.get(pre + ".norm1.body.bias"),
                               with_bias_ln)
        a, cache = hmha(p, pre + ".attn", n1, cache, sw, tap, forced_order)
        x = x + a
    n2 = channel_layernorm(x, p[pre + ".norm2.body.weight"], p.get(pre + ".norm2.body.bias"),
                           with_bias_ln)
    x = x + gdfn(p, pre + ".ffn", n2)
    if tap:
        tap(pre + ".out", x.clone())
    return x, cache


def downsample(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Downsample (:454-471): conv3x3 C->C/2 (no bias) then pixel-unshuffle(2)."""
    return pixel_unshuffle2(conv3x3(x, p[pre + ".body.0.weight"]))


def upsample(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Upsample (:474-491): conv3x3 C->2C (no bias) then pixel-shuffle(2)."""
    return pixel_shuffle2(conv3x3(x, p[pre + ".body.0.weight"]))


def hint_forward(state: dict, x: torch.Tensor, cfg: Optional[dict] = None,
                 qv_cache: Optional[torch.Tensor] = None,
                 sw: Optional[HintSwitches] = None, tap: Tap = None,
                 forced_orders: Optional[dict] = None):
    """HINT.execute (:657-706) -> (y, qv_cache_out).

    A call with B > 1 is the reference's literal batched call: the regroup
    statistics and the cache use batch means (:245, :282).  `forced_orders`
    ({block name: channel order}) replaces the argsort of the named blocks (a
    checker's tool: the argsort is discontinuous in its input).
    """
    cfg = cfg or default_config()
    sw = sw or HintSwitches()
    p = as_params(state)
    wb = cfg.get("ln_type", "WithBias") != "BiasFree"
    x = x.float()
    cache = qv_cache

    def run_stage(name, n, t, is_att):
        nonlocal cache
        for i in range(n):
            fo = None if forced_orders is None else forced_orders.get(f"{name}.{i}")
            t, cache = transformer_block(p, f"{name}.{i}", t, cache, is_att, wb, sw, tap, fo)
        return t

    plan = {s[0]: s for s in stage_plan(cfg)}
    emb = conv3x3(x, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"))
    e1 = run_stage("encoder_level1", plan["encoder_level1"][1], emb, False)
    e2 = run_stage("encoder_level2", plan["encoder_level2"][1], downsample(p, "down1_2", e1), False)
    e3 = run_stage("encoder_level3", plan["encoder_level3"][1], downsample(p, "down2_3", e2), False)
    lat = run_stage("latent", plan["latent"][1], downsample(p, "down3_4", e3), True)

    d3 = torch.cat([upsample(p, "up4_3", lat), e3], dim=1)
    d3 = conv1x1(d3, p["reduce_chan_level3.weight"], p.get("reduce_chan_level3.bias"))
    d3 = run_stage("decoder_level3", plan["decoder_level3"][1], d3, True)

    d2 = torch.cat([upsample(p, "up3_2", d3), e2], dim=1)
    d2 = conv1x1(d2, p["reduce_chan_level2.weight"], p.get("reduce_chan_level2.bias"))
    d2 = run_stage("decoder_level2", plan["decoder_level2"][1], d2, True)

    d1 = torch.cat([upsample(p, "up2_1", d2), e1], dim=1)
    d1 = run_stage("decoder_level1", plan["decoder_level1"][1], d1, True)
    d1 = run_stage("refinement", plan["refinement"][1], d1, True)

    if cfg.get("dual_pixel_task", False):
        d1 = d1 + conv1x1(emb, p["skip_conv.weight"], p.get("skip_conv.bias"))
        y = conv3x3(d1, p["output.weight"], p.get("output.bias"))
    else:
        y = conv3x3(d1, p["output.weight"], p.get("output.bias")) + x
    return y, cache


def hint_forward_per_image(state: dict, x: torch.Tensor, cfg: Optional[dict] = None,
                           sw: Optional[HintSwitches] = None):
    """B independent cold-start (qv_cache=None) B=1 reference calls, stacked."""
    ys, caches = [], []
    for i in range(x.shape[0]):
        y, c = hint_forward(state, x[i:i + 1], cfg, None, sw)
        ys.append(y)
        caches.append(c)
    return torch.cat(ys, 0), torch.cat(caches, 0)
