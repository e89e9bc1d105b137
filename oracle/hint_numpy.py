"""Second, independent restatement of the HINT forward in NumPy float64 (test infrastructure).

Written from the reference's ``basicsr/models/archs/hint_arch.py`` without sharing code with
``hint_oracle.py`` (no torch ops: convolutions are shift-and-add, the bilinear resize and the
adaptive pooling are literal loops), so that the two can check each other.  Slow: for tiny
configurations only.  Parity unpinned (no reference vectors exist).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def _conv3x3(x, w, groups=1):
    """x [B,Cin,H,W], w [Cout,Cin/groups,3,3], zero pad 1."""
    b, cin, h, ww = x.shape
    cout = w.shape[0]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    y = np.zeros((b, cout, h, ww))
    if groups == 1:
        for dy in range(3):
            for dx in range(3):
                y += np.einsum("oc,bchw->bohw", w[:, :, dy, dx], xp[:, :, dy:dy + h, dx:dx + ww])
    else:  # depthwise: groups == cin == cout
        for dy in range(3):
            for dx in range(3):
                y += w[None, :, 0, dy, dx, None, None] * xp[:, :, dy:dy + h, dx:dx + ww]
    return y


def _conv1x1(x, w):
    return np.einsum("oc,bchw->bohw", w[:, :, 0, 0], x)


def _conv1d(x, w, bias):
    return np.einsum("oi,bin->bon", w[:, :, 0], x) + bias[None, :, None]


def _ln(x, w, bias):
    """hint_arch.py:110-112,129-132 (per pixel over channels; BiasFree when bias is None)."""
    mu = x.mean(axis=1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    if bias is None:
        return x / np.sqrt(var + 1e-5) * w[None, :, None, None]
    return (x - mu) / np.sqrt(var + 1e-5) * w[None, :, None, None] + bias[None, :, None, None]


def _unshuffle(x):
    """hint_arch.py:48-56 as an explicit index map."""
    b, c, h, w = x.shape
    y = np.zeros((b, 4 * c, h // 2, w // 2))
    for i in range(2):
        for j in range(2):
            y[:, 2 * i + j::4] = x[:, :, i::2, j::2]
    return y


def _shuffle(x):
    b, c4, h, w = x.shape
    y = np.zeros((b, c4 // 4, 2 * h, 2 * w))
    for i in range(2):
        for j in range(2):
            y[:, :, i::2, j::2] = x[:, 2 * i + j::4]
    return y


def _adaptive_pool(x, n_out):
    """Literal loops of AdaptiveAvgPool2d.execute (hint_arch.py:69-95) on a [H,W] matrix."""
    h_in, w_in = x.shape
    out = np.zeros((n_out, n_out))
    for i in range(n_out):
        hs, he = math.floor(i * h_in / n_out), math.ceil((i + 1) * h_in / n_out)
        for j in range(n_out):
            ws, we = math.floor(j * w_in / n_out), math.ceil((j + 1) * w_in / n_out)
            out[i, j] = x[hs:he, ws:we].mean()
    return out


def _bilinear_up(m, n_out):
    """bilinear, align_corners=False (half-pixel centres, source index clamped at 0)."""
    n_in = m.shape[0]
    scale = n_in / n_out
    out = np.zeros((n_out, n_out))
    for y in range(n_out):
        sy = max(scale * (y + 0.5) - 0.5, 0.0)
        y0 = min(int(math.floor(sy)), n_in - 1)
        y1 = min(y0 + 1, n_in - 1)
        ly = sy - y0
        for x in range(n_out):
            sx = max(scale * (x + 0.5) - 0.5, 0.0)
            x0 = min(int(math.floor(sx)), n_in - 1)
            x1 = min(x0 + 1, n_in - 1)
            lx = sx - x0
            out[y, x] = ((1 - ly) * ((1 - lx) * m[y0, x0] + lx * m[y0, x1])
                         + ly * ((1 - lx) * m[y1, x0] + lx * m[y1, x1]))
    return out


def _attention(p, pre, x, cache):
    """Attention.execute (hint_arch.py:345-405)."""
    b, c, h, w = x.shape
    qkv = _conv3x3(_conv1x1(x, p[pre + ".qkv.weight"]), p[pre + ".qkv_dwconv.weight"], groups=3 * c)
    q, k, v = [t.reshape(b, c, h * w) for t in np.split(qkv, 3, axis=1)]
    feat = q.mean(axis=0)
    xm = feat - feat.mean(axis=1, keepdims=True)
    cov = xm @ xm.T / (feat.shape[1] - 1)
    d = np.sqrt(np.diag(cov))
    sim = (cov / np.outer(d, d)).mean(axis=1)
    order = np.argsort(-sim, kind="stable")
    q, k, v = q[:, order], k[:, order], v[:, order]
    sizes = [int(r / 8 * c) for r in (1, 2, 2, 3)]
    att, tc = [], []
    s0 = 0
    for g, cg in enumerate(sizes):
        qn = q[:, s0:s0 + cg] / np.sqrt(np.maximum((q[:, s0:s0 + cg] ** 2).sum(-1, keepdims=True), 1e-30))
        kn = k[:, s0:s0 + cg] / np.sqrt(np.maximum((k[:, s0:s0 + cg] ** 2).sum(-1, keepdims=True), 1e-30))
        a = np.einsum("bil,bjl->bij", qn, kn) * p[pre + ".temperature"][g, 0, 0]
        a = np.exp(a - a.max(-1, keepdims=True))
        att.append(a / a.sum(-1, keepdims=True))
        tc.append(qn + kn)
        s0 += cg
    tc = np.concatenate(tc, axis=1)
    if cache is not None and cache.shape[-1] != c:
        cache = _adaptive_pool(cache[0], c)[None]
    outs = []
    s0 = 0
    for g, cg in enumerate(sizes):
        if cache is not None:
            m = f"{pre}.inter_modulator{g + 1}"
            pooled = _adaptive_pool(cache[0], cg)[None]
            z = _conv1d(pooled + att[g], p[m + ".conv_width.weight"], p[m + ".conv_width.bias"])
            y = att[g] * z[:, :cg] + z[:, cg:]
            y = y * _gelu(_conv1d(y, p[m + ".gating_conv.weight"], p[m + ".gating_conv.bias"]))
            att[g] = y + att[g]
        outs.append(att[g] @ v[:, s0:s0 + cg])
        s0 += cg
    upd = np.zeros((c, c))
    for a in att:
        cg = a.shape[-1]
        # the reference floors an fp32 product; evaluate the product in fp32 too so that values a
        # hair below an integer fall on the same side
        up = _bilinear_up(a.mean(axis=0), c).astype(np.float32)
        upd += np.floor((up * np.float32(cg)) / np.float32(c))
    cache = upd[None] * 0.9 if cache is None else cache * 0.9 + upd[None] * (1 - 0.9)
    out = np.concatenate(outs, axis=1)
    im = pre + ".intra_modulator"
    s = tc + out
    gt = _gelu(_conv1d(s, p[im + ".gating_conv.weight"], p[im + ".gating_conv.bias"])) * s
    out = out + _conv1d(_conv1d(gt, p[im + ".down.weight"], p[im + ".down.bias"]),
                        p[im + ".up.weight"], p[im + ".up.bias"])
    return _conv1x1(out.reshape(b, c, h, w), p[pre + ".project_out.weight"]), cache, order


def _block(p, pre, x, cache, att, orders):
    if att:
        a, cache, order = _attention(p, pre + ".attn", _ln(x, p[pre + ".norm1.body.weight"],
                                                           p.get(pre + ".norm1.body.bias")), cache)
        orders[pre] = order
        x = x + a
    n = _ln(x, p[pre + ".norm2.body.weight"], p.get(pre + ".norm2.body.bias"))
    t = _conv3x3(_conv1x1(n, p[pre + ".ffn.project_in.weight"]), p[pre + ".ffn.dwconv.weight"],
                 groups=p[pre + ".ffn.dwconv.weight"].shape[0])
    x1, x2 = np.split(t, 2, axis=1)
    return x + _conv1x1(_gelu(x1) * x2, p[pre + ".ffn.project_out.weight"]), cache


def hint_forward_numpy(state: dict, x: np.ndarray, cfg: dict, cache=None):
    """HINT.execute (hint_arch.py:657-706) for bias=False configurations. Returns (y, cache, orders)."""
    assert not cfg.get("bias", False) and not cfg.get("dual_pixel_task", False)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in state.items()}
    x = np.asarray(x, dtype=np.float64)
    nb = cfg["num_blocks"]
    orders = {}

    def stage(name, n, t, att):
        nonlocal cache
        for i in range(n):
            t, cache = _block(p, f"{name}.{i}", t, cache, att, orders)
        return t

    e1 = stage("encoder_level1", nb[0], _conv3x3(x, p["patch_embed.proj.weight"]), False)
    e2 = stage("encoder_level2", nb[1], _unshuffle(_conv3x3(e1, p["down1_2.body.0.weight"])), False)
    e3 = stage("encoder_level3", nb[2], _unshuffle(_conv3x3(e2, p["down2_3.body.0.weight"])), False)
    lat = stage("latent", nb[1], _unshuffle(_conv3x3(e3, p["down3_4.body.0.weight"])), True)
    d3 = np.concatenate([_shuffle(_conv3x3(lat, p["up4_3.body.0.weight"])), e3], axis=1)
    d3 = stage("decoder_level3", nb[2], _conv1x1(d3, p["reduce_chan_level3.weight"]), True)
    d2 = np.concatenate([_shuffle(_conv3x3(d3, p["up3_2.body.0.weight"])), e2], axis=1)
    d2 = stage("decoder_level2", nb[1], _conv1x1(d2, p["reduce_chan_level2.weight"]), True)
    d1 = np.concatenate([_shuffle(_conv3x3(d2, p["up2_1.body.0.weight"])), e1], axis=1)
    d1 = stage("decoder_level1", nb[0], d1, True)
    d1 = stage("refinement", cfg["num_refinement_blocks"], d1, True)
    return _conv3x3(d1, p["output.weight"]) + x, cache, orders
