"""Oracle for the ASTv2 forward (test infrastructure, parity unpinned).

Restates ``basicsr/models/archs/astv2_arch.py`` of the reference in torch-CPU fp32; line numbers
in the docstrings refer to that file.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .common import (ASSUMPTIONS, Params, as_params, channel_layernorm, conv1x1, conv3x3, gelu, pixel_shuffle2,
                     pixel_unshuffle2)

FFN_GROUPS = 12   # FeedForward.groups, :82
Tap = Optional[Callable[[str, torch.Tensor], None]]


def default_config() -> dict:
    """network_g block of Desnowing/Options/Desnowing_ASTv2.yml:47-63."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[2, 4, 6], num_refinement_blocks=2,
                heads=[2, 4, 8], ffn_expansion_factor=4, bias=False, ln_type="WithBias",
                dual_pixel_task=False)


def stage_plan(cfg: dict):
    """(name, n_blocks, dim multiplier, head index, is_att) in execution order (:320-426)."""
    nb = cfg["num_blocks"]
    return [
        ("encoder_level1", nb[0], 1, 0, False),
        ("encoder_level2", nb[1], 2, 1, False),
        ("encoder_level3", nb[2], 4, 2, False),
        ("decoder_level3", nb[2], 4, 2, True),
        ("decoder_level2", nb[1], 2, 1, True),
        ("decoder_level1", nb[0], 2, 0, True),
        ("refinement", cfg["num_refinement_blocks"], 2, 0, True),
    ]


def l2_normalize_rows(x: torch.Tensor, eps: float = None) -> torch.Tensor:
    """jt.normalize(x, dim=-1) (:174-175): x / sqrt(max(sum x^2, eps)), eps = ASSUMPTIONS.normalize_eps."""
    eps = ASSUMPTIONS.normalize_eps if eps is None else eps
    n2 = (x * x).sum(dim=-1, keepdim=True)
    return x / torch.sqrt(torch.clamp(n2, min=eps))


def partial_conv(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """:111-121 — view as [B*12, C/12, H, W]; the first C/24 channels of every one of the 12
    groups go through ONE shared depthwise 3x3 (weight [C/24,1,3,3]); the rest pass through."""
    b, c, h, w = x.shape
    g = c // FFN_GROUPS
    dc = c // (2 * FFN_GROUPS)
    xg = x.reshape(b * FFN_GROUPS, g, h, w)
    x1 = conv3x3(xg[:, :dc], p[pre + ".partial_conv3.weight"], None, groups=dc)
    return torch.cat([x1, xg[:, dc:]], dim=1).reshape(b, c, h, w)


def feed_forward(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward.execute (:110-131): partial conv, 1x1 up, dw3x3 on the first half, product
    with the un-convolved second half (no GELU), 1x1 down."""
    x = partial_conv(p, pre, x)
    t = conv1x1(x, p[pre + ".project_in.weight"], p.get(pre + ".project_in.bias"))
    half = t.shape[1] // 2
    x1 = conv3x3(t[:, :half], p[pre + ".dwconv.weight"], p.get(pre + ".dwconv.bias"), groups=half)
    return conv1x1(x1 * t[:, half:], p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, temperature: torch.Tensor,
                   ca_w: torch.Tensor, ca_b: Optional[torch.Tensor], heads: int):
    """The part of Attention.execute between the two convolutions (:170-202).  q, k, v [b, C, L];
    returns (attn [b, heads, ch, ch], out [b, C, L])."""
    b, c, L = q.shape
    ch = c // heads
    q, k, v = [t.reshape(b, heads, ch, L) for t in (q, k, v)]
    q, k = l2_normalize_rows(q), l2_normalize_rows(k)
    attn = (q @ k.transpose(-2, -1)) * temperature.reshape(1, heads, 1, 1)
    attn0 = attn.softmax(dim=-1)
    attn1 = torch.relu(attn) ** 2
    attn1 = gelu(attn1) * attn1                                   # [b, head, ch(row), ch(col)]
    # att_ca is a 1x1 Conv2d over the ROW axis: the tensor is viewed as (N=b, C=ch_row, H=head, W=ch_col)
    z = conv1x1(attn1.permute(0, 2, 1, 3), ca_w, ca_b)
    z = z.permute(0, 2, 3, 1)                                     # [b, head, ch_col, 2ch]
    scale, shift = z[..., :ch], z[..., ch:]
    attn = attn0 * (1 + scale) + shift                            # positional: scale[r, c] = Z[c-th out][r-th col]
    return attn, (attn @ v).reshape(b, c, L)


def attention(p: Params, pre: str, x: torch.Tensor, heads: int, tap: Tap = None) -> torch.Tensor:
    """Attention.execute (:158-206)."""
    b, c, h, w = x.shape
    qkv = conv3x3(conv1x1(x, p[pre + ".qkv.weight"], p.get(pre + ".qkv.bias")),
                  p[pre + ".qkv_dwconv.weight"], p.get(pre + ".qkv_dwconv.bias"), groups=3 * c)
    q, k, v = [t.reshape(b, c, h * w) for t in qkv.split(c, dim=1)]
    attn, out = attention_core(q, k, v, p[pre + ".temperature"], p[pre + ".att_ca.weight"], p.get(pre + ".att_ca.bias"), heads)
    if tap:
        tap(pre + ".attn", attn.clone())
    return conv1x1(out.reshape(b, c, h, w), p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def transformer_block(p: Params, pre: str, x, heads: int, is_att: bool, wb: bool, tap: Tap = None):
    """TransformerBlock.execute (:220-225)."""
    if is_att:
        n1 = channel_layernorm(x, p[pre + ".norm1.body.weight"], p.get(pre + ".norm1.body.bias"), wb)
        x = x + attention(p, pre + ".attn", n1, heads, tap)
    n2 = channel_layernorm(x, p[pre + ".norm2.body.weight"], p.get(pre + ".norm2.body.bias"), wb)
    x = x + feed_forward(p, pre + ".ffn", n2)
    if tap:
        tap(pre + ".out", x.clone())
    return x


def astv2_forward(state: dict, x: torch.Tensor, cfg: Optional[dict] = None, tap: Tap = None) -> torch.Tensor:
    """ASTv2.execute (:438-469)."""
    cfg = cfg or default_config()
    p = as_params(state)
    wb = cfg.get("ln_type", "WithBias") != "BiasFree"
    heads = cfg["heads"]
    x = x.float()
    plan = {s[0]: s for s in stage_plan(cfg)}

    def run(name, t):
        _, n, _, hi, is_att = plan[name]
        for i in range(n):
            t = transformer_block(p, f"{name}.{i}", t, heads[hi], is_att, wb, tap)
        return t

    emb = conv3x3(x, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"))
    e1 = run("encoder_level1", emb)
    e2 = run("encoder_level2", pixel_unshuffle2(conv3x3(e1, p["down1_2.body.0.weight"])))
    e3 = run("encoder_level3", pixel_unshuffle2(conv3x3(e2, p["down2_3.body.0.weight"])))
    d3 = run("decoder_level3", e3)
    d2 = torch.cat([pixel_shuffle2(conv3x3(d3, p["up3_2.body.0.weight"])), e2], dim=1)
    d2 = run("decoder_level2", conv1x1(d2, p["reduce_chan_level2.weight"], p.get("reduce_chan_level2.bias")))
    d1 = torch.cat([pixel_shuffle2(conv3x3(d2, p["up2_1.body.0.weight"])), e1], dim=1)
    d1 = run("refinement", run("decoder_level1", d1))
    if cfg.get("dual_pixel_task", False):
        d1 = d1 + conv1x1(emb, p["skip_conv.weight"], p.get("skip_conv.bias"))
        return conv3x3(d1, p["output.weight"], p.get("output.bias"))
    return conv3x3(d1, p["output.weight"], p.get("output.bias")) + x
