"""Oracle for the AST forward (test infrastructure, parity unpinned).

Restates ``basicsr/models/archs/ast_arch.py`` of the reference in torch-CPU fp32: sparse window attention
wherever attention is on (the class hard-codes that), token MLP by ``token_mlp`` (frfn as shipped, leff, ffn / mlp);
line numbers in the docstrings refer to that file.  Tokens are kept as ``[B, L, C]`` like the
reference, with ``H = W = int(sqrt(L))``.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from .common import Params, as_params, gelu, leaky_relu, token_layernorm

Tap = Optional[Callable[[str, torch.Tensor], None]]
LAYERS = ["encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
          "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3"]


def default_config() -> dict:
    """Enhancement/Options/AST_LOL_v2_synthetic.yml:45-47 -> AST(token_mlp="frfn") + ctor defaults (:941-963)."""
    return dict(img_size=256, in_chans=3, dd_in=3, embed_dim=32, depths=[2] * 9,
                num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4.0, qkv_bias=True,
                token_mlp="frfn", shift_flag=True)


def layer_plan(cfg: dict):
    """(name, dim multiplier, resolution divisor, depth, heads, att) per BasicASTLayer (:998-1198)."""
    d, h = cfg["depths"], cfg["num_heads"]
    mult = [1, 2, 4, 8, 16, 16, 8, 4, 2]
    div = [1, 2, 4, 8, 16, 8, 4, 2, 1]
    return [(LAYERS[i], mult[i], div[i], d[i], h[i], i >= 4) for i in range(9)]


def block_window(cfg: dict, res_div: int, index: int):
    """(win_size, shift_size) of block `index` (:712-715, :886): the window shrinks to the layer's
    NOMINAL resolution (img_size / divisor) when that is <= win_size, and then never shifts."""
    win = cfg["win_size"]
    shift = win // 2 if (cfg.get("shift_flag", True) and index % 2 == 1) else 0
    nominal = cfg["img_size"] // res_div
    if nominal <= win:
        return nominal, 0
    return win, shift


def bmm_4d(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """:33-49 — both operands are FLATTENED to [B*H, ., .] in their own memory order."""
    B, H, N, C = a.shape
    a_ = a.reshape(B * H, N, C)
    if tuple(b.shape[-2:]) == (N, C):
        b_ = b.reshape(B * H, N, C).transpose(1, 2)
    elif tuple(b.shape[-2:]) == (C, N):
        b_ = b.reshape(B * H, C, N)
    else:
        raise ValueError("unexpected shape")
    return torch.bmm(a_, b_).reshape(B, H, N, -1)


def relative_position_index(win: int) -> torch.Tensor:
    """:248-257 — idx[p, q] = (ph - qh + win - 1) * (2 win - 1) + (pw - qw + win - 1)."""
    coords = torch.stack(torch.meshgrid(torch.arange(win), torch.arange(win), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += win - 1
    rel[:, :, 1] += win - 1
    rel[:, :, 0] *= 2 * win - 1
    return rel.sum(-1)


def window_partition(x: torch.Tensor, win: int) -> torch.Tensor:
    """:533-534 — [B,H,W,C] -> [B*nW, win, win, C], windows row-major over (H/win, W/win)."""
    B, H, W, C = x.shape
    return x.reshape(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, C)


def window_reverse(w: torch.Tensor, win: int, H: int, W: int) -> torch.Tensor:
    """:540-553."""
    B = int(w.shape[0] / (H * W / win / win))
    return w.reshape(B, H // win, W // win, win, win, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_attn_mask(H: int, W: int, win: int, shift: int) -> torch.Tensor:
    """:786-805 — region ids 0..8 on the H x W grid, partitioned; differing ids -> -100."""
    m = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -win), slice(-win, -shift), slice(-shift, None)):
        for ws in (slice(0, -win), slice(-win, -shift), slice(-shift, None)):
            m[:, hs, ws, :] = cnt
            cnt += 1
    mw = window_partition(m, win).reshape(-1, win * win)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def sparse_attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, table: torch.Tensor, w: torch.Tensor,
                          win: int, mask: Optional[torch.Tensor], tap: Tap = None, tap_name: str = "") -> torch.Tensor:
    """The body of WindowAttention_sparse.execute between the projections (:283-316).  q (already scaled), k, v:
    [B_, heads, N, d]; returns [B_, N, heads * d]."""
    B_, heads, N, d = q.shape
    # k is handed over as [heads, B_, d, N]; bmm_4d flattens it in THAT order while q is flattened
    # as [B_, heads, ...]: score (b, h) pairs q[b, h] with k[b', h'], h' * B_ + b' = b * heads + h.
    attn = bmm_4d(q, k.transpose(0, 1).transpose(2, 3))
    idx = relative_position_index(win).reshape(-1)
    bias = table[idx].reshape(win * win, win * win, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.reshape(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape(-1, heads, N, N)
    attn0 = torch.softmax(attn, dim=-1)
    attn1 = torch.relu(attn) ** 2
    w_soft = torch.exp(w) / torch.sum(torch.exp(w))
    attn = attn0 * w_soft[0] + attn1 * w_soft[1]
    if tap:
        tap(tap_name, attn.clone())
    return (attn @ v).transpose(1, 2).reshape(B_, N, heads * d)


def window_attention_sparse(p: Params, pre: str, x: torch.Tensor, heads: int, win: int,
                            mask: Optional[torch.Tensor], tap: Tap = None) -> torch.Tensor:
    """WindowAttention_sparse.execute (:275-318) with LinearProjection (:105-117). x: [B_, N, C]."""
    B_, N, C = x.shape
    d = C // heads
    q = F.linear(x, p[pre + ".qkv.to_q.weight"], p.get(pre + ".qkv.to_q.bias"))
    kv = F.linear(x, p[pre + ".qkv.to_kv.weight"], p.get(pre + ".qkv.to_kv.bias"))
    q = q.reshape(B_, N, 1, heads, d).permute(2, 0, 3, 1, 4)[0]          # [B_, heads, N, d]
    kv = kv.reshape(B_, N, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    q = q * (d ** -0.5)
    out = sparse_attention_core(q, k, v, p[pre + ".relative_position_bias_table"], p[pre + ".w"], win, mask, tap,
                                pre + ".attn")
    return F.linear(out, p[pre + ".proj.weight"], p.get(pre + ".proj.bias"))


def frfn(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """FRFN.execute (:476-513). x: [B, L, C]."""
    b, hw, c = x.shape
    hh = int(math.sqrt(hw))
    dc = c // 4
    t = x.reshape(b, hh, hh, c).permute(0, 3, 1, 2)
    t = torch.cat([F.conv2d(t[:, :dc], p[pre + ".partial_conv3.weight"], None, padding=1), t[:, dc:]], dim=1)
    t = t.permute(0, 2, 3, 1).reshape(b, hw, c)
    t = gelu(F.linear(t, p[pre + ".linear1.0.weight"], p[pre + ".linear1.0.bias"]))
    x1, x2 = t.chunk(2, dim=-1)
    h = x1.shape[-1]
    x1 = x1.reshape(b, hh, hh, h).permute(0, 3, 1, 2)
    x1 = gelu(F.conv2d(x1, p[pre + ".dwconv.0.weight"], p[pre + ".dwconv.0.bias"], padding=1, groups=h))
    x1 = x1.permute(0, 2, 3, 1).reshape(b, hw, h)
    return F.linear(x1 * x2, p[pre + ".linear2.0.weight"], p[pre + ".linear2.0.bias"])


def leff(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """LeFF.execute (:425-450): linear1 + GELU -> NCHW -> depthwise 3x3 (bias) + GELU -> tokens -> linear2;
    `eca` is nn.Identity (:421)."""
    b, hw, c = x.shape
    hh = int(math.sqrt(hw))
    t = gelu(F.linear(x, p[pre + ".linear1.0.weight"], p[pre + ".linear1.0.bias"]))
    h = t.shape[-1]
    t = t.reshape(b, hh, hh, h).permute(0, 3, 1, 2)
    t = gelu(F.conv2d(t, p[pre + ".dwconv.0.weight"], p[pre + ".dwconv.0.bias"], padding=1, groups=h))
    t = t.permute(0, 2, 3, 1).reshape(b, hw, h)
    return F.linear(t, p[pre + ".linear2.0.weight"], p[pre + ".linear2.0.bias"])


def mlp(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Mlp.execute (:396-402): fc1 -> GELU -> fc2 (dropout is the identity in eval)."""
    return F.linear(gelu(F.linear(x, p[pre + ".fc1.weight"], p[pre + ".fc1.bias"])), p[pre + ".fc2.weight"],
                    p[pre + ".fc2.bias"])


def token_mlp(kind: str):
    """TransformerBlock.__init__ (:741-757): 'ffn' / 'mlp' -> Mlp, 'leff' -> LeFF, 'frfn' -> FRFN."""
    if kind in ("ffn", "mlp"):
        return mlp
    if kind == "leff":
        return leff
    if kind == "frfn":
        return frfn
    raise Exception("FFN error!")


def _drop_path(t: torch.Tensor, scales: Optional[dict], key: str) -> torch.Tensor:
    """DropPath in training mode (:52-70): the branch times a per-sample factor, 0 or 1 / keep_prob.  The factors are
    given (the reference draws them from Jittor's generator); None or a missing key = the branch is kept."""
    if not scales or key not in scales:
        return t
    f = torch.as_tensor(scales[key], dtype=t.dtype).reshape((t.shape[0],) + (1,) * (t.ndim - 1))
    return t * f


def transformer_block(p: Params, pre: str, x: torch.Tensor, heads: int, win: int, shift: int,
                      att: bool, tap: Tap = None, ffn=frfn, drop_scales: Optional[dict] = None) -> torch.Tensor:
    """TransformerBlock.execute (:768-842), `mask` argument None (as in every script)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    if att:
        mask = shift_attn_mask(H, W, win, shift) if shift > 0 else None
        t = token_layernorm(x, p[pre + ".norm1.weight"], p[pre + ".norm1.bias"]).reshape(B, H, W, C)
        if shift > 0:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        wins = window_partition(t, win).reshape(-1, win * win, C)
        a = window_attention_sparse(p, pre + ".attn", wins, heads, win, mask, tap)
        t = window_reverse(a.reshape(-1, win, win, C), win, H, W)
        if shift > 0:
            t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
        x = x + _drop_path(t.reshape(B, H * W, C), drop_scales, pre + ".attn")
    x = x + _drop_path(ffn(p, pre + ".mlp", token_layernorm(x, p[pre + ".norm2.weight"], p[pre + ".norm2.bias"])),
                       drop_scales, pre + ".mlp")
    if tap:
        tap(pre + ".out", x.clone())
    return x


def _to_img(t: torch.Tensor) -> torch.Tensor:
    B, L, C = t.shape
    s = int(math.sqrt(L))
    return t.transpose(1, 2).reshape(B, C, s, s)


def _to_tokens(t: torch.Tensor) -> torch.Tensor:
    return t.flatten(2).transpose(1, 2)


def drop_path_rates(cfg: dict) -> dict:
    """Per-block drop-path rate (:981-983, :1010-1190): linspace(0, rate, number of encoder blocks) over the four
    encoder layers, `rate` for the bottleneck, the encoder list reversed over the four decoder layers."""
    d = cfg["depths"]
    rate = float(cfg.get("drop_path_rate", 0.1))
    n_enc = sum(d[:4])
    enc = [float(v) for v in torch.linspace(0, rate, n_enc)] if n_enc > 1 else [0.0] * n_enc
    dec = enc[::-1]
    out, e, k = {}, 0, 0
    for li in range(9):
        for j in range(d[li]):
            if li < 4:
                r = enc[e]; e += 1
            elif li == 4:
                r = rate
            else:
                r = dec[k]; k += 1
            out[f"{LAYERS[li]}.blocks.{j}"] = r
    return out


def ast_forward(state: dict, x: torch.Tensor, cfg: Optional[dict] = None, tap: Tap = None,
                drop_scales: Optional[dict] = None) -> torch.Tensor:
    """AST.execute (:1220-1256).  drop_scales: training-mode DropPath factors per branch ("<block>.attn" /
    "<block>.mlp" -> [B]); None = evaluation mode."""
    cfg = cfg or default_config()
    p = as_params(state)
    x = x.float()
    plan = layer_plan(cfg)
    ffn = token_mlp(cfg.get("token_mlp", "leff"))

    def layer(i, t):
        name, _, div, depth, heads, att = plan[i]
        for j in range(depth):
            win, shift = block_window(cfg, div, j)
            t = transformer_block(p, f"{name}.blocks.{j}", t, heads, win, shift, att, tap, ffn, drop_scales)
        return t

    def down(i, t):
        return _to_tokens(F.conv2d(_to_img(t), p[f"dowsample_{i}.conv.0.weight"], p[f"dowsample_{i}.conv.0.bias"],
                                   stride=2, padding=1))

    def up(i, t):
        return _to_tokens(F.conv_transpose2d(_to_img(t), p[f"upsample_{i}.deconv.0.weight"],
                                             p[f"upsample_{i}.deconv.0.bias"], stride=2))

    y = _to_tokens(leaky_relu(F.conv2d(x, p["input_proj.proj.0.weight"], p["input_proj.proj.0.bias"], padding=1)))
    c0 = layer(0, y)
    c1 = layer(1, down(0, c0))
    c2 = layer(2, down(1, c1))
    c3 = layer(3, down(2, c2))
    c4 = layer(4, down(3, c3))
    d0 = layer(5, torch.cat([up(0, c4), c3], -1))
    d1 = layer(6, torch.cat([up(1, d0), c2], -1))
    d2 = layer(7, torch.cat([up(2, d1), c1], -1))
    d3 = layer(8, torch.cat([up(3, d2), c0], -1))
    out = F.conv2d(_to_img(d3), p["output_proj.proj.0.weight"], p["output_proj.proj.0.bias"], padding=1)
    return x + out if cfg.get("dd_in", 3) == 3 else out
