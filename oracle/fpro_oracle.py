"""Oracle for the FPro forward (test infrastructure, parity unpinned).

Restates ``basicsr/models/archs/fpro_arch.py`` of the reference in torch-CPU fp32, literally
(including the full 2-D FFT round trip of the low-frequency prompt); line numbers in the docstrings
refer to that file.  Semantics of the Jittor runtime that cannot be checked here (Jittor is not
installable) are named switches of FproSwitches:
  * bicubic resize = torch's (cubic convolution, A = -0.75, half-pixel centres, no antialias);
  * ComplexNumber.ifft2 carries the 1 / (H W) factor (numpy convention), fft2 none.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from .common import (Params, as_params, channel_layernorm, conv1x1, conv3x3, gelu, pixel_shuffle2,
                     pixel_unshuffle2)

Tap = Optional[Callable[[str, torch.Tensor], None]]
WIN = 8            # PromptModule: win_size=8 (:464-469)
PROMPT_HEADS = 2   # PromptModule: num_heads=2
FILTER_GROUPS = 8  # dynamic_filter_channel: group=8 (:340)


@dataclass
class FproSwitches:
    ifft_scaled: bool = True       # ifft2 divides by H*W
    bicubic: str = "torch"         # only value built


def default_config() -> dict:
    """network_g block of Dehaze/Options/RealDehazing_FPro.yml:45-61."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[2, 3, 6], num_refinement_blocks=2,
                heads=[2, 4, 8], ffn_expansion_factor=3, bias=False, ln_type="WithBias", dual_pixel_task=False)


def stage_plan(cfg: dict):
    """(name, n_blocks, dim multiplier, head index, is_att) in execution order (:563-676)."""
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


def prompt_plan(cfg: dict):
    """(name, dim multiplier, nominal input_resolution) (:621-676)."""
    return [("prompt_d3", 4, 64), ("prompt_d2", 2, 128), ("prompt_d1", 2, 256), ("prompt_r", 2, 256)]


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """:10-12 — the module-level monkey patch: x / (||x||_2 + 1e-12)."""
    return x / (torch.linalg.vector_norm(x, dim=-1, keepdim=True) + 1e-12)


def feed_forward(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward.execute (:109-114)."""
    t = conv1x1(x, p[pre + ".project_in.weight"], p.get(pre + ".project_in.bias"))
    t = conv3x3(t, p[pre + ".dwconv.weight"], p.get(pre + ".dwconv.bias"), groups=t.shape[1])
    x1, x2 = t.chunk(2, dim=1)
    return conv1x1(gelu(x1) * x2, p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, temperature: torch.Tensor, heads: int):
    """MDTA between the two convolutions (:134-142).  q, k, v [b, C, L]; returns (attn [b, heads, ch, ch] after the
    softmax, out [b, C, L])."""
    b, c, L = q.shape
    ch = c // heads
    q, k, v = [t.reshape(b, heads, ch, L) for t in (q, k, v)]
    q, k = normalize_rows(q), normalize_rows(k)
    attn = ((q @ k.transpose(-2, -1)) * temperature.reshape(1, heads, 1, 1)).softmax(dim=-1)
    return attn, (attn @ v).reshape(b, c, L)


def attention(p: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """Attention.execute (:128-144): MDTA."""
    b, c, h, w = x.shape
    qkv = conv3x3(conv1x1(x, p[pre + ".qkv.weight"], p.get(pre + ".qkv.bias")),
                  p[pre + ".qkv_dwconv.weight"], p.get(pre + ".qkv_dwconv.bias"), groups=3 * c)
    q, k, v = [t.reshape(b, c, h * w) for t in qkv.split(c, dim=1)]
    _, out = attention_core(q, k, v, p[pre + ".temperature"], heads)
    return conv1x1(out.reshape(b, c, h, w), p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def transformer_block(p: Params, pre: str, x, heads: int, is_att: bool, wb: bool):
    """TransformerBlock.execute (:157-161)."""
    if is_att:
        n1 = channel_layernorm(x, p[pre + ".norm1.body.weight"], p.get(pre + ".norm1.body.bias"), wb)
        x = x + attention(p, pre + ".attn", n1, heads)
    n2 = channel_layernorm(x, p[pre + ".norm2.body.weight"], p.get(pre + ".norm2.body.bias"), wb)
    return x + feed_forward(p, pre + ".ffn", n2)


def dynamic_filter(p: Params, pre: str, x: torch.Tensor):
    """dynamic_filter_channel.execute (:355-375): a per-image, per-group 3x3 low-pass (softmax over
    the 9 taps) applied with reflection padding; high = x - low."""
    n, c, h, w = x.shape
    g, k2 = FILTER_GROUPS, 9
    pooled = x.mean(dim=(2, 3), keepdim=True)
    f = conv1x1(pooled, p[pre + ".conv.weight"])
    f = f * torch.sigmoid(conv1x1(f, p[pre + ".conv_gate.weight"]))
    f = (f - p[pre + ".bn.running_mean"].reshape(1, -1, 1, 1)) / torch.sqrt(
        p[pre + ".bn.running_var"].reshape(1, -1, 1, 1) + 1e-5)
    f = f * p[pre + ".bn.weight"].reshape(1, -1, 1, 1) + p[pre + ".bn.bias"].reshape(1, -1, 1, 1)
    unf = F.unfold(F.pad(x, (1, 1, 1, 1), mode="reflect"), kernel_size=3)          # [n, c*9, h*w]
    unf = unf.reshape(n, g, c // g, k2, h * w)
    filt = f.reshape(n, g, k2, 1).unsqueeze(2).softmax(dim=-2)                       # [n, g, 1, 9, 1]
    low = (unf * filt).sum(dim=3).reshape(n, c, h, w)
    return low, x - low


def fspg_high(p: Params, pre: str, ffm: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """frequenctSpecificPromptGenetator, flag_highF=True (:404-411)."""
    ffm = F.interpolate(ffm, size=(H, W), mode="bilinear", align_corners=False)
    c = ffm.shape[1]
    t = F.conv2d(ffm, p[pre + ".body.0.weight"], p[pre + ".body.0.bias"], padding=(0, 1), groups=c)
    t = F.conv2d(t, p[pre + ".body.1.weight"], p[pre + ".body.1.bias"], padding=(1, 0), groups=c)
    wr = F.interpolate(p[pre + ".weight"], size=(H, W), mode="bilinear", align_corners=False)
    return gelu(t) * ffm * wr


def fspg_low(p: Params, pre: str, ffm: torch.Tensor, H: int, W: int, sw: FproSwitches) -> torch.Tensor:
    """frequenctSpecificPromptGenetator, flag_highF=False (:413-447)."""
    ffm = F.interpolate(ffm, size=(H, W), mode="bicubic", align_corners=False)
    y = torch.fft.fft2(ffm)                                   # per sample, over the last two axes
    y_f = torch.cat([y.real, y.imag], dim=1)
    y_att = gelu(conv1x1(y_f, p[pre + ".body.0.weight"], p[pre + ".body.0.bias"]))
    y_f = y_f * y_att
    yr, yi = y_f.chunk(2, dim=1)
    cw = p[pre + ".complex_weight"]
    wr = F.interpolate(cw[..., 0], size=(H, W), mode="bilinear", align_corners=False)
    wi = F.interpolate(cw[..., 1], size=(H, W), mode="bilinear", align_corners=False)
    prod = torch.complex(yr * wr - yi * wi, yr * wi + yi * wr)
    out = torch.fft.ifft2(prod, norm="backward" if sw.ifft_scaled else "forward")
    if not sw.ifft_scaled:
        out = out * (H * W) / (H * W)   # "forward" already leaves the sum unscaled
    return out.real.contiguous()


def window_partition(x: torch.Tensor, win: int) -> torch.Tensor:
    """:174-191 (dilation 1) — [B,H,W,C] -> [B*nW, win, win, C]."""
    B, H, W, C = x.shape
    return x.reshape(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, C)


def window_reverse(w: torch.Tensor, win: int, H: int, W: int) -> torch.Tensor:
    """:194-209."""
    B = int(w.shape[0] / (H * W / win / win))
    return w.reshape(B, H // win, W // win, win, win, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def cross_window_core(q: torch.Tensor, kv: torch.Tensor, heads: int) -> torch.Tensor:
    """:322-332 — q [B_, N, C] (after to_q), kv [B_, N, 2C] (after to_kv) per window -> [B_, N, C]: softmax attention
    per head, the product transposed to [.., d, N] and then RESHAPED (not permuted back) to [B_, N, C]."""
    B_, N, c = q.shape
    hd = c // heads
    q = q.reshape(B_, N, 1, heads, hd).permute(2, 0, 3, 1, 4)[0] * (hd ** -0.5)
    kv = kv.reshape(B_, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    return (attn @ v).transpose(-2, -1).reshape(B_, N, c)         # the literal flatten of :331-332


def high_fusion(p: Params, pre: str, feat: torch.Tensor, kvf: torch.Tensor) -> torch.Tensor:
    """highFrequencyPromptFusion.execute (:309-336): 8x8-window cross attention, q from the feature,
    k / v from the (depthwise-convolved) prompt; the product is transposed to [.., d, N] and then
    RESHAPED (not permuted back) to [B_, N, C] (:331-332)."""
    b, c, h, w = feat.shape
    c2 = kvf.shape[1]
    kvf = conv3x3(kvf, p[pre + ".kv_dwconv.weight"], p.get(pre + ".kv_dwconv.bias"), groups=c2)
    qw = window_partition(feat.permute(0, 2, 3, 1), WIN).reshape(-1, WIN * WIN, c)
    kw = window_partition(kvf.permute(0, 2, 3, 1), WIN).reshape(-1, WIN * WIN, c2)
    q = F.linear(qw, p[pre + ".to_q.to_q.weight"], p.get(pre + ".to_q.to_q.bias"))
    kv = F.linear(kw, p[pre + ".to_kv.to_kv.weight"], p.get(pre + ".to_kv.to_kv.bias"))
    out = cross_window_core(q, kv, PROMPT_HEADS)
    out = F.linear(out, p[pre + ".project_out.weight"], p[pre + ".project_out.bias"])
    out = window_reverse(out.reshape(-1, WIN, WIN, c), WIN, h, w)
    return out.permute(0, 3, 1, 2).contiguous()


def low_fusion(p: Params, pre: str, feat: torch.Tensor, prompt: torch.Tensor) -> torch.Tensor:
    """lowFrequencyPromptFusion.execute (:222-246): keys / values come from the globally pooled
    prompt (one token), so the softmax runs over a single key."""
    b, c1, h, w = feat.shape
    heads, ch = PROMPT_HEADS, c1 // PROMPT_HEADS
    query = conv1x1(feat, p[pre + ".q.weight"], p.get(pre + ".q.bias")).reshape(b, h * w, heads, ch).permute(0, 2, 1, 3)
    pooled = prompt.mean(dim=(2, 3), keepdim=True)
    kv = conv1x1(pooled, p[pre + ".kv.weight"], p.get(pre + ".kv.bias"))
    kv = kv.reshape(b, 2 * c1, -1).transpose(1, 2).reshape(b, -1, 2, heads, ch).permute(2, 0, 3, 1, 4)
    key, value = kv[0], kv[1]
    attn = (query @ key.transpose(-2, -1)) * p[pre + ".temperature"].reshape(1, heads, 1, 1)
    out = attn.softmax(dim=-1) @ value                                            # [b, head, hw, ch]
    out = out.permute(0, 1, 3, 2).reshape(b, c1, h, w)                            # b (head c) h w
    return conv1x1(out, p[pre + ".project_out.weight"], p.get(pre + ".project_out.bias"))


def prompt_module(p: Params, pre: str, low: torch.Tensor, high: torch.Tensor, x: torch.Tensor,
                  sw: FproSwitches, tap: Tap = None) -> torch.Tensor:
    """PromptModule.execute (:471-478)."""
    _, _, h, w = x.shape
    y_h = fspg_high(p, pre + ".FSPG_high", high, h, w)
    y_l = fspg_low(p, pre + ".FSPG_low", low, h, w, sw)
    if tap:
        tap(pre + ".y_h", y_h.clone())
        tap(pre + ".y_l_mean", y_l.mean(dim=(2, 3)))
    y_h = high_fusion(p, pre + ".modulator_hi", x, y_h)
    y_l = low_fusion(p, pre + ".modulator_lo", x, y_l)
    if tap:
        tap(pre + ".hi", y_h.clone())
        tap(pre + ".lo", y_l.clone())
    return conv1x1(torch.cat([y_h, y_l], dim=1), p[pre + ".simple_Fusion.weight"], p[pre + ".simple_Fusion.bias"])


def fpro_forward(state: dict, x: torch.Tensor, cfg: Optional[dict] = None, sw: Optional[FproSwitches] = None,
                 tap: Tap = None) -> torch.Tensor:
    """FPro.execute (:684-718)."""
    cfg = cfg or default_config()
    sw = sw or FproSwitches()
    p = as_params(state)
    wb = cfg.get("ln_type", "WithBias") != "BiasFree"
    heads = cfg["heads"]
    x = x.float()
    plan = {s[0]: s for s in stage_plan(cfg)}

    def run(name, t):
        _, n, _, hi, is_att = plan[name]
        for i in range(n):
            t = transformer_block(p, f"{name}.{i}", t, heads[hi], is_att, wb)
        if tap:
            tap(name, t.clone())
        return t

    emb = conv3x3(x, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"))
    e1 = run("encoder_level1", emb)
    e2 = run("encoder_level2", pixel_unshuffle2(conv3x3(e1, p["down1_2.body.0.weight"])))
    e3 = run("encoder_level3", pixel_unshuffle2(conv3x3(e2, p["down2_3.body.0.weight"])))
    d3 = run("decoder_level3", e3)
    low, high = dynamic_filter(p, "splitFre.dyna_channel", emb)
    if tap:
        tap("low", low.clone())
    d3 = prompt_module(p, "prompt_d3", low, high, d3, sw, tap) + d3
    t = torch.cat([pixel_shuffle2(conv3x3(d3, p["up3_2.body.0.weight"])), e2], dim=1)
    t = conv1x1(t, p["reduce_chan_level2.weight"], p.get("reduce_chan_level2.bias"))
    d2 = run("decoder_level2", t)
    d2 = prompt_module(p, "prompt_d2", low, high, d2, sw, tap) + d2
    t = torch.cat([pixel_shuffle2(conv3x3(d2, p["up2_1.body.0.weight"])), e1], dim=1)
    d1 = run("decoder_level1", t)
    d1 = prompt_module(p, "prompt_d1", low, high, d1, sw, tap) + d1
    d1 = run("refinement", d1)
    d1 = prompt_module(p, "prompt_r", low, high, d1, sw, tap) + d1
    if cfg.get("dual_pixel_task", False):
        d1 = d1 + conv1x1(emb, p["skip_conv.weight"], p.get("skip_conv.bias"))
        return conv3x3(d1, p["output.weight"], p.get("output.bias"))
    return conv3x3(d1, p["output.weight"], p.get("output.bias")) + x
