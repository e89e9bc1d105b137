"""CPU tests of the oracle: golden-fixture regression, two independent restatements against each
other, and the index/formula ops on their own.  (Parity unpinned: the fixtures come from the
oracle itself — see tests/make_golden.py.)"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import common as oc
from oracle import hint_oracle as ho
from oracle.hint_numpy import _adaptive_pool, _bilinear_up, hint_forward_numpy
from oracle.weights import make_hint_state, synthetic_images
from tests.make_golden import HINT_SMALL, weights_digest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,cfg", [("hint_small_48x64", HINT_SMALL), ("hint_full_32x32", None),
                                      ("hint_small_modgain_32x48", HINT_SMALL)])
def test_oracle_reproduces_golden(name, cfg):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = cfg or ho.default_config()
    sd = make_hint_state(cfg, seed=int(g["seed"]), gain=0.6, mod_gain=float(g["mod_gain"]))
    assert weights_digest(sd) == str(g["weights_sha256"]), "weight generator changed: regenerate the fixtures"
    taps = {}
    y1, c1 = ho.hint_forward(sd, torch.from_numpy(g["x1"]), cfg, None, tap=lambda n, t: taps.__setitem__(n, t))
    perms = np.concatenate([taps[n].numpy() for n in taps if n.endswith(".attn.perm")])
    assert np.array_equal(perms, g["perms1"])
    assert np.abs(y1.numpy() - g["y1"]).max() < 2e-5
    assert np.abs(c1.numpy() - g["cache1"]).max() < 1e-6
    y2, c2 = ho.hint_forward(sd, torch.from_numpy(g["x2"]), cfg, torch.from_numpy(g["cache1"]))
    assert np.abs(y2.numpy() - g["y2"]).max() < 2e-5
    assert np.mean(np.abs(c2.numpy() - g["cache2"]) > 1e-6) < 1e-3


TINY = dict(inp_channels=3, out_channels=3, dim=8, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1,
            heads=[1, 1, 1, 1], ffn_expansion_factor=2.66, bias=False, ln_type="WithBias", dual_pixel_task=False)


@pytest.mark.parametrize("ln_type,mod_gain", [("WithBias", 1.0), ("BiasFree", 1.0), ("WithBias", 3.0)])
def test_two_restatements_agree(ln_type, mod_gain):
    """torch-fp32 oracle vs the NumPy-float64 transcription: same channel orders, outputs within
    fp32 noise, cold call and warm call."""
    cfg = dict(TINY, ln_type=ln_type)
    sd = make_hint_state(cfg, seed=2, gain=0.6, mod_gain=mod_gain)
    x = synthetic_images(2, 16, 24, seed=4, sigma=50)
    taps = {}
    y, c = ho.hint_forward(sd, torch.from_numpy(x), cfg, None, tap=lambda n, t: taps.__setitem__(n, t))
    yn, cn, orders = hint_forward_numpy(sd, x, cfg, None)
    for name, order in orders.items():
        assert np.array_equal(order, taps[name + ".attn.perm"].numpy()), name
    assert np.abs(y.numpy() - yn).max() < 5e-5
    assert np.mean(np.abs(c.numpy() - cn) > 1e-5) < 5e-3
    y2, c2 = ho.hint_forward(sd, torch.from_numpy(x), cfg, torch.from_numpy(cn.astype(np.float32)))
    yn2, cn2, _ = hint_forward_numpy(sd, x, cfg, cn)
    assert np.abs(y2.numpy() - yn2).max() < 5e-5
    assert np.mean(np.abs(c2.numpy() - cn2) > 1e-5) < 5e-3


def test_pixel_shuffle_maps():
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    u = oc.pixel_unshuffle2(x)
    for c in range(3):
        for i in range(2):
            for j in range(2):
                assert torch.equal(u[:, 4 * c + 2 * i + j], x[:, c, i::2, j::2])
    assert torch.equal(oc.pixel_shuffle2(u), x)
    assert torch.equal(u, F.pixel_unshuffle(x, 2))
    assert torch.equal(oc.pixel_shuffle2(u), F.pixel_shuffle(u, 2))


@pytest.mark.parametrize("n_in,n_out", [(384, 192), (192, 96), (96, 384), (384, 144), (96, 36), (96, 12), (192, 72), (7, 5)])
def test_adaptive_pool_bounds_and_values(n_in, n_out):
    s, e = oc.adaptive_pool_bounds(n_in, n_out)
    for i in range(n_out):     # the reference's float expressions (hint_arch.py:80-81)
        assert s[i] == math.floor(i * n_in / n_out)
        assert e[i] == math.ceil((i + 1) * n_in / n_out)
        assert e[i] > s[i]
    x = torch.randn(1, n_in, n_in, generator=torch.Generator().manual_seed(0))
    got = oc.adaptive_avg_pool_sq(x, n_out)[0].numpy()
    if n_in * n_out <= 96 * 36:
        assert np.abs(got - _adaptive_pool(x[0].double().numpy(), n_out)).max() < 1e-6
    assert np.abs(got - F.adaptive_avg_pool2d(x[None], n_out)[0, 0].numpy()).max() < 1e-5


def test_bilinear_matches_independent_formula():
    m = torch.randn(12, 12, generator=torch.Generator().manual_seed(1))
    up = F.interpolate(m[None, None], size=(96, 96), mode="bilinear", align_corners=False)[0, 0]
    assert np.abs(up.numpy() - _bilinear_up(m.double().numpy(), 96)).max() < 1e-6


def test_layer_cache_floor_semantics():
    sw = ho.HintSwitches()
    att = [torch.full((1, 2, 2), 5.0), torch.full((1, 4, 4), -0.3), torch.full((1, 4, 4), 0.2), torch.full((1, 6, 6), 2.9)]
    got = ho.layer_cache(att, 16, sw)
    # floor(5*2/16)=0, floor(-0.3*4/16)=-1, floor(0.2*4/16)=0, floor(2.9*6/16)=1
    assert torch.equal(got, torch.zeros(1, 16, 16))
    assert torch.equal(ho.layer_cache(att, 16, ho.HintSwitches(floordiv="trunc")), torch.ones(1, 16, 16))


def test_group_sizes_and_stage_plan():
    assert ho.group_sizes(96) == [12, 24, 24, 36]
    assert ho.group_sizes(384) == [48, 96, 96, 144]
    plan = ho.stage_plan(ho.default_config())
    assert [(n, k) for n, k, _, _ in plan] == [("encoder_level1", 4), ("encoder_level2", 6), ("encoder_level3", 6),
                                               ("latent", 6), ("decoder_level3", 6), ("decoder_level2", 6),
                                               ("decoder_level1", 4), ("refinement", 4)]


def test_batch_coupling_is_visible():
    """A literal B=2 call differs from two B=1 calls (batch-mean statistics), hence two batch modes."""
    sd = make_hint_state(TINY, seed=2, gain=0.6)
    x = torch.from_numpy(synthetic_images(2, 16, 16, seed=9, sigma=50))
    y_c, _ = ho.hint_forward(sd, x, TINY)
    y_p, c_p = ho.hint_forward_per_image(sd, x, TINY)
    assert c_p.shape[0] == 2
    assert float((y_c - y_p).abs().max()) > 1e-4
    y0, _ = ho.hint_forward(sd, x[:1], TINY)
    assert torch.equal(y0, y_p[:1])


# ---- ASTv2 / AST oracles ---------------------------------------------------------------------------
def test_astv2_and_ast_oracles_reproduce_golden():
    from oracle import ast_oracle, astv2_oracle
    from oracle.weights import make_ast_state, make_astv2_state
    from tests.make_golden import AST_SHALLOW, ASTV2_SMALL
    g = np.load(os.path.join(GOLD, "astv2_small_40x56.npz"))
    sd = make_astv2_state(ASTV2_SMALL, seed=int(g["seed"]), gain=0.6)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = astv2_oracle.astv2_forward(sd, torch.from_numpy(g["x"]), ASTV2_SMALL).numpy()
    assert np.abs(y - g["y"]).max() < 2e-5
    g = np.load(os.path.join(GOLD, "ast_shallow_128.npz"))
    sd = make_ast_state(AST_SHALLOW, seed=int(g["seed"]), gain=0.5)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = ast_oracle.ast_forward(sd, torch.from_numpy(g["x"]), AST_SHALLOW).numpy()
    assert np.abs(y - g["y"]).max() < 2e-5


def test_ast_relative_position_index_and_shift_mask():
    from oracle import ast_oracle as so
    idx = so.relative_position_index(8).numpy()
    for p in range(64):           # literal loop form of ast_arch.py:248-257
        for q in range(64):
            assert idx[p, q] == (p // 8 - q // 8 + 7) * 15 + (p % 8 - q % 8 + 7)
    assert idx.min() == 0 and idx.max() == 224
    H = W = 16
    m = so.shift_attn_mask(H, W, 8, 4).numpy()        # [nW, 64, 64]
    region = np.zeros((H, W), np.int64)
    for y in range(H):
        for x in range(W):
            ry = 0 if y < H - 8 else (1 if y < H - 4 else 2)
            rx = 0 if x < W - 8 else (1 if x < W - 4 else 2)
            region[y, x] = 3 * ry + rx
    for w in range(4):
        wy, wx = divmod(w, 2)
        r = region[wy * 8:wy * 8 + 8, wx * 8:wx * 8 + 8].reshape(-1)
        want = np.where(r[None, :] != r[:, None], -100.0, 0.0)
        assert np.array_equal(m[w], want)


def test_ast_bmm_4d_pairs_q_with_another_windows_k():
    """The flattening of ast_arch.py:33-49: q is laid out [window, head], k [head, window], so score
    block j = b * heads + h multiplies q[b, h] with k[j // B_, j % B_]."""
    from oracle import ast_oracle as so
    rng = torch.Generator().manual_seed(0)
    B_, heads, N, D = 6, 2, 4, 3
    q = torch.randn(B_, heads, N, D, generator=rng)
    k = torch.randn(heads, B_, N, D, generator=rng)       # the reference's k after its permute
    got = so.bmm_4d(q, k)
    for b in range(B_):
        for h in range(heads):
            j = b * heads + h
            assert torch.allclose(got[b, h], q[b, h] @ k[j // B_, j % B_].T, atol=1e-6)


def test_ast_window_partition_round_trip():
    from oracle import ast_oracle as so
    x = torch.arange(2 * 16 * 24 * 3, dtype=torch.float32).reshape(2, 16, 24, 3)
    w = so.window_partition(x, 8)
    assert w.shape == (2 * 2 * 3, 8, 8, 3)
    assert torch.equal(w[4], x[0, 8:16, 8:16])            # window (1, 1) of image 0
    assert torch.equal(so.window_reverse(w, 8, 16, 24), x)


# ---- FPro oracle ------------------------------------------------------------------------------------
def test_fpro_oracle_reproduces_golden():
    from oracle import fpro_oracle
    from oracle.weights import make_fpro_state
    from tests.make_golden import FPRO_SMALL
    g = np.load(os.path.join(GOLD, "fpro_small_64x96.npz"))
    sd = make_fpro_state(FPRO_SMALL, seed=int(g["seed"]), gain=0.6)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = fpro_oracle.fpro_forward(sd, torch.from_numpy(g["x"]), FPRO_SMALL).numpy()
    assert np.abs(y - g["y"]).max() < 2e-5


def test_fpro_dynamic_filter_is_a_normalised_low_pass():
    """The nine taps of every (image, group) are a softmax: a constant image passes through the
    low branch unchanged and leaves a zero high branch (fpro_arch.py:355-375)."""
    from oracle import fpro_oracle as fo
    from oracle.weights import make_fpro_state
    from tests.make_golden import FPRO_SMALL
    p = {k: torch.from_numpy(v) for k, v in make_fpro_state(FPRO_SMALL, seed=3).items()}
    x = torch.full((2, 16, 12, 20), 0.7)
    low, high = fo.dynamic_filter(p, "splitFre.dyna_channel", x)
    assert float((low - x).abs().max()) < 1e-6 and float(high.abs().max()) < 1e-6
    # literal loop form on a random image: reflection padding, per-group taps
    x = torch.randn(1, 16, 6, 7, generator=torch.Generator().manual_seed(0))
    low, _ = fo.dynamic_filter(p, "splitFre.dyna_channel", x)
    pooled = x.mean(dim=(2, 3))[0]
    f = p["splitFre.dyna_channel.conv.weight"][:, :, 0, 0] @ pooled
    f = f * torch.sigmoid(p["splitFre.dyna_channel.conv_gate.weight"][:, :, 0, 0] @ f)
    f = (f - p["splitFre.dyna_channel.bn.running_mean"]) / torch.sqrt(p["splitFre.dyna_channel.bn.running_var"] + 1e-5)
    f = f * p["splitFre.dyna_channel.bn.weight"] + p["splitFre.dyna_channel.bn.bias"]
    taps = f.reshape(8, 9).softmax(dim=1)
    refl = lambda i, n: -i if i < 0 else (2 * (n - 1) - i if i >= n else i)
    for c in (0, 5, 15):
        for (yy, xx) in ((0, 0), (5, 6), (2, 3)):
            want = sum(float(taps[c // 2, (dy + 1) * 3 + dx + 1]) * float(x[0, c, refl(yy + dy, 6), refl(xx + dx, 7)])
                       for dy in (-1, 0, 1) for dx in (-1, 0, 1))
            assert abs(float(low[0, c, yy, xx]) - want) < 1e-5


def test_fpro_high_fusion_flatten_is_the_literal_reshape():
    """(attn @ v).transpose(-2, -1).reshape(B_, N, C) (fpro_arch.py:331-332): flat element
    (head * hd + d) * N + n of a window lands on token f // C, channel f % C."""
    from oracle import fpro_oracle as fo
    B_, heads, N, hd = 3, 2, 64, 4
    C = heads * hd
    o = torch.arange(B_ * heads * N * hd, dtype=torch.float32).reshape(B_, heads, N, hd)
    flat = o.transpose(-2, -1).reshape(B_, N, C)
    for (b, h, n, d) in ((0, 0, 0, 0), (1, 1, 17, 3), (2, 0, 63, 2)):
        f = (h * hd + d) * N + n
        assert float(flat[b, f // C, f % C]) == float(o[b, h, n, d])


def test_fpro_low_prompt_mean_is_the_dc_term():
    """mean over pixels of ifft2(Y).real == Re(Y[0, 0]) / (H W): what lets the device path skip the FFT."""
    from oracle import fpro_oracle as fo
    from oracle.weights import make_fpro_state
    from tests.make_golden import FPRO_SMALL
    p = {k: torch.from_numpy(v) for k, v in make_fpro_state(FPRO_SMALL, seed=4).items()}
    low = torch.rand(2, 16, 64, 96, generator=torch.Generator().manual_seed(1))
    H, W = 32, 48
    y = fo.fspg_low(p, "prompt_d2.FSPG_low", low, H, W, fo.FproSwitches())
    r = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)
    s = r.sum(dim=(2, 3))                                                    # [B, 16] = Re Y[0,0], Im = 0
    yf = torch.cat([s, torch.zeros_like(s)], dim=1)
    a = fo.gelu(yf @ p["prompt_d2.FSPG_low.body.0.weight"][:, :, 0, 0].T + p["prompt_d2.FSPG_low.body.0.bias"])
    re = (yf * a)[:, :16]
    wr = torch.nn.functional.interpolate(p["prompt_d2.FSPG_low.complex_weight"][..., 0], size=(H, W), mode="bilinear",
                                         align_corners=False)[0, :, 0, 0]
    want = re * wr / (H * W)
    got = y.mean(dim=(2, 3))
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("field,value,arch", [
    ("gelu", "tanh", "ASTv2"), ("gelu", "tanh", "AST"), ("variance", "unbiased", "ASTv2"), ("variance", "unbiased", "AST"),
    ("layernorm_eps", 1e-2, "HINT"), ("leaky_relu_slope", 0.2, "AST"), ("pixel_shuffle_order", "grouped", "ASTv2"),
    ("normalize_eps", 1e3, "ASTv2"),
])
def test_named_jittor_assumptions_are_live_switches(field, value, arch):
    """Every switchable entry of JittorAssumptions changes the oracle's output when flipped (so the name is
    wired to the arithmetic it describes) and the default set reproduces the unflipped result exactly."""
    import dataclasses
    from oracle import ast_oracle, astv2_oracle, common
    from oracle.weights import make_ast_state, make_astv2_state, synthetic_images
    if arch == "AST":
        cfg = dict(ast_oracle.default_config(), depths=[1] * 9)
        sd, run, x = make_ast_state(cfg, seed=2, gain=0.5), ast_oracle.ast_forward, synthetic_images(1, 128, 128, seed=3)
    elif arch == "ASTv2":
        cfg = dict(astv2_oracle.default_config(), dim=24, num_blocks=[1, 1, 1], num_refinement_blocks=1)
        sd, run, x = make_astv2_state(cfg, seed=2, gain=0.5), astv2_oracle.astv2_forward, synthetic_images(1, 16, 24, seed=3)
    else:
        cfg = dict(ho.default_config(), dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
        sd, x = make_hint_state(cfg, seed=2, gain=0.6), synthetic_images(1, 16, 24, seed=3)
        run = lambda s, t, c: ho.hint_forward(s, t, c, None)[0]
    base = run(sd, torch.from_numpy(x), cfg).numpy()
    saved = dataclasses.replace(common.ASSUMPTIONS)
    try:
        setattr(common.ASSUMPTIONS, field, value)
        flipped = run(sd, torch.from_numpy(x), cfg).numpy()
    finally:
        for f in dataclasses.fields(saved):
            setattr(common.ASSUMPTIONS, f.name, getattr(saved, f.name))
    again = run(sd, torch.from_numpy(x), cfg).numpy()
    assert np.array_equal(base, again)
    assert np.abs(flipped - base).max() > 1e-6, f"{field}={value!r} did not reach the {arch} oracle"


def _erf_gelu64(x):
    return 0.5 * x * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def test_astv2_attention_mix_against_a_float64_loop_transcription():
    """Attention.execute astv2_arch.py:174-200 written a second time, independently of oracle/astv2_oracle.py: NumPy
    float64, explicit loops over (batch, head, row, column), the `att_ca` 1x1 convolution as a sum over the ROW index of
    the squared-ReLU map and its output read back transposed - the index formula
        Z[o, j] = sum_i W[o, i] R[i, j];  scale[r, c] = Z[c, r];  shift[r, c] = Z[ch + c, r]
    that the three permutes of :187-196 amount to - and, beside it, the permutes applied literally with np.transpose.
    Both must equal the torch oracle's attention matrix and output."""
    from oracle import astv2_oracle as ao
    rng = np.random.default_rng(7)
    b, heads, ch, L = 2, 3, 5, 37
    c = heads * ch
    q, k, v = (rng.standard_normal((b, c, L)) for _ in range(3))
    temp = rng.uniform(0.5, 2.0, (heads, 1, 1))
    W = rng.standard_normal((2 * ch, ch)) * 0.4
    bias = rng.standard_normal(2 * ch) * 0.1
    t32 = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    attn_o, out_o = ao.attention_core(t32(q), t32(k), t32(v), t32(temp), t32(W.reshape(2 * ch, ch, 1, 1)), t32(bias), heads)
    q4, k4, v4 = (a.astype(np.float32).astype(np.float64).reshape(b, heads, ch, L) for a in (q, k, v))
    W, bias, temp = (a.astype(np.float32).astype(np.float64) for a in (W, bias, temp))
    qn = q4 / np.sqrt((q4 ** 2).sum(-1, keepdims=True))
    kn = k4 / np.sqrt((k4 ** 2).sum(-1, keepdims=True))
    A = np.zeros((b, heads, ch, ch))
    for n in range(b):
        for h in range(heads):
            for r in range(ch):
                for col in range(ch):
                    A[n, h, r, col] = float(np.dot(qn[n, h, r], kn[n, h, col])) * temp[h, 0, 0]
    S = np.exp(A - A.max(-1, keepdims=True))
    S /= S.sum(-1, keepdims=True)
    R = np.maximum(A, 0.0) ** 2
    R = _erf_gelu64(R) * R
    # (1) the index formula, loops only
    mixed = np.zeros_like(A)
    for n in range(b):
        for h in range(heads):
            Z = np.zeros((2 * ch, ch))
            for o in range(2 * ch):
                for j in range(ch):
                    Z[o, j] = sum(W[o, i] * R[n, h, i, j] for i in range(ch)) + bias[o]
            for r in range(ch):
                for col in range(ch):
                    mixed[n, h, r, col] = S[n, h, r, col] * (1.0 + Z[col, r]) + Z[ch + col, r]
    # (2) the reference's own sequence of permutes, literally
    a1 = np.transpose(R, (0, 2, 1, 3))                                   # (b, ch_row, head, ch_col): NCHW of the conv
    x_att = np.einsum("oi,nihj->nohj", W, a1) + bias.reshape(1, -1, 1, 1)  # (b, 2ch, head, ch_col)
    x_att = np.transpose(x_att, (0, 2, 3, 1))                            # (b, head, ch_col, 2ch)
    literal = S * (1.0 + x_att[..., :ch]) + x_att[..., ch:]
    assert np.abs(mixed - literal).max() < 1e-12
    out = (mixed @ v4).reshape(b, c, L)
    assert np.abs(attn_o.numpy() - mixed).max() < 2e-5
    assert np.abs(out_o.numpy() - out).max() < 2e-4 * max(1.0, np.abs(out).max())
    # the transposition is visible: reading Z without it gives another matrix
    wrong = S * (1.0 + np.transpose(x_att[..., :ch], (0, 1, 3, 2))) + np.transpose(x_att[..., ch:], (0, 1, 3, 2))
    assert np.abs(wrong - mixed).max() > 1e-2


def test_fpro_dynamic_filter_against_a_float64_loop_transcription():
    """dynamic_filter_channel.execute fpro_arch.py:355-375 a second time, NumPy float64 with explicit loops over every
    output element: global average pool -> 1x1 -> sigmoid gate -> eval-mode BatchNorm -> softmax over the 9 taps of each
    of the 8 groups -> reflection-padded 3x3 unfold, channel c taking the taps of group c // (C / 8) - the grouping the
    reshape (n, group, c // group, 9, h w) of :363-365 expresses.  Every element of low and high must match the oracle."""
    from oracle import fpro_oracle as fo
    from oracle.weights import make_fpro_state
    from tests.make_golden import FPRO_SMALL
    sd = make_fpro_state(FPRO_SMALL, seed=5)
    pre = "splitFre.dyna_channel"
    rng = np.random.default_rng(11)
    n, c, h, w = 2, int(FPRO_SMALL["dim"]), 7, 9
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    low_o, high_o = fo.dynamic_filter(p, pre, torch.from_numpy(x))
    f64 = lambda key: np.asarray(sd[pre + key], np.float64)
    Wc, Wg = f64(".conv.weight")[:, :, 0, 0], f64(".conv_gate.weight")[:, :, 0, 0]
    mean, var, gam, bet = f64(".bn.running_mean"), f64(".bn.running_var"), f64(".bn.weight"), f64(".bn.bias")
    G, per = 8, c // 8
    xd = x.astype(np.float64)
    low = np.zeros_like(xd)
    refl = lambda i, m: -i if i < 0 else (2 * (m - 1) - i if i >= m else i)
    for b in range(n):
        pooled = np.array([xd[b, ch].sum() / (h * w) for ch in range(c)])
        f = Wc @ pooled
        f = f * (1.0 / (1.0 + np.exp(-(Wg @ f))))
        f = (f - mean) / np.sqrt(var + 1e-5) * gam + bet
        taps = f.reshape(G, 9)
        taps = np.exp(taps - taps.max(1, keepdims=True))
        taps /= taps.sum(1, keepdims=True)
        for ch in range(c):
            g = ch // per
            for y in range(h):
                for xx in range(w):
                    acc = 0.0
                    for ky in range(3):
                        for kx in range(3):
                            acc += taps[g, ky * 3 + kx] * xd[b, ch, refl(y + ky - 1, h), refl(xx + kx - 1, w)]
                    low[b, ch, y, xx] = acc
    assert np.abs(low_o.numpy() - low).max() < 2e-6
    assert np.abs(high_o.numpy() - (xd - low)).max() < 2e-6
