"""Synthetic inputs and weights for tests, the benchmark and the smoke run.

No checkpoint or image ships with the reference, so everything is driven by seeded random weights
in the reference's state-dict key / shape layout (one generator per `network_g` class) and seeded
random images.  Conv / linear kernels are N(0, gain^2 / fan_in) so activations stay O(1) through
the whole depth; every per-channel parameter (LayerNorm weight and bias, temperature, conv biases)
is randomised so that a kernel which applies one to the wrong channel is caught.

The constructor defaults and stage plans below restate the reference's constructors (file:line in
each docstring); they carry no arithmetic of the forward pass.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

# ---- HINT (basicsr/models/archs/hint_arch.py) ----
def hint_default_config() -> dict:
    """network_g block of Denoising/Options/GaussianColorDenoising_HINT.yml:51-62."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                num_refinement_blocks=4, heads=[8, 8, 8, 8], ffn_expansion_factor=2.66,
                bias=False, ln_type="WithBias", dual_pixel_task=False)


def hint_stage_plan(cfg: dict):
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


# ---- ASTv2 (basicsr/models/archs/astv2_arch.py) ----
def astv2_default_config() -> dict:
    """network_g block of Desnowing/Options/Desnowing_ASTv2.yml:47-63."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[2, 4, 6], num_refinement_blocks=2,
                heads=[2, 4, 8], ffn_expansion_factor=4, bias=False, ln_type="WithBias",
                dual_pixel_task=False)


def astv2_stage_plan(cfg: dict):
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


# ---- AST (basicsr/models/archs/ast_arch.py) ----
AST_LAYERS = ['encoderlayer_0', 'encoderlayer_1', 'encoderlayer_2', 'encoderlayer_3', 'conv', 'decoderlayer_0', 'decoderlayer_1', 'decoderlayer_2', 'decoderlayer_3']


def ast_default_config() -> dict:
    """Enhancement/Options/AST_LOL_v2_synthetic.yml:45-47 -> AST(token_mlp="frfn") + ctor defaults (:941-963)."""
    return dict(img_size=256, in_chans=3, dd_in=3, embed_dim=32, depths=[2] * 9,
                num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4.0, qkv_bias=True,
                token_mlp="frfn", shift_flag=True)


def ast_layer_plan(cfg: dict):
    """(name, dim multiplier, resolution divisor, depth, heads, att) per BasicASTLayer (:998-1198)."""
    d, h = cfg["depths"], cfg["num_heads"]
    mult = [1, 2, 4, 8, 16, 16, 8, 4, 2]
    div = [1, 2, 4, 8, 16, 8, 4, 2, 1]
    return [(AST_LAYERS[i], mult[i], div[i], d[i], h[i], i >= 4) for i in range(9)]


def ast_block_window(cfg: dict, res_div: int, index: int):
    """(win_size, shift_size) of block `index` (:712-715, :886): the window shrinks to the layer's
    NOMINAL resolution (img_size / divisor) when that is <= win_size, and then never shifts."""
    win = cfg["win_size"]
    shift = win // 2 if (cfg.get("shift_flag", True) and index % 2 == 1) else 0
    nominal = cfg["img_size"] // res_div
    if nominal <= win:
        return nominal, 0
    return win, shift


def ast_relative_position_index(win: int) -> np.ndarray:
    """ast_arch.py:248-257 — idx[p, q] = (ph - qh + win - 1) * (2 win - 1) + (pw - qw + win - 1)."""
    ph, pw = np.divmod(np.arange(win * win), win)
    return ((ph[:, None] - ph[None, :] + win - 1) * (2 * win - 1) + (pw[:, None] - pw[None, :] + win - 1)).astype(np.int32)


# ---- FPro (basicsr/models/archs/fpro_arch.py) ----
FPRO_FILTER_GROUPS = 8  # dynamic_filter_channel: group=8 (fpro_arch.py:340)


def fpro_default_config() -> dict:
    """network_g block of Dehaze/Options/RealDehazing_FPro.yml:45-61."""
    return dict(inp_channels=3, out_channels=3, dim=48, num_blocks=[2, 3, 6], num_refinement_blocks=2,
                heads=[2, 4, 8], ffn_expansion_factor=3, bias=False, ln_type="WithBias", dual_pixel_task=False)


def fpro_stage_plan(cfg: dict):
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


def fpro_prompt_plan(cfg: dict):
    """(name, dim multiplier, nominal input_resolution) (:621-676)."""
    return [("prompt_d3", 4, 64), ("prompt_d2", 2, 128), ("prompt_d1", 2, 256), ("prompt_r", 2, 256)]


class _Gen:
    def __init__(self, seed: int, gain: float = 1.0):
        self.rng = np.random.default_rng(seed)
        self.gain = gain
        self.sd = OrderedDict()

    def kernel(self, key, shape, fan_in=None):
        if fan_in is None:
            fan_in = int(np.prod(shape[1:]))
        std = self.gain / np.sqrt(max(fan_in, 1))
        self.sd[key] = (self.rng.standard_normal(shape) * std).astype(np.float32)

    def vec(self, key, n, mean=0.0, std=0.1):
        self.sd[key] = (mean + std * self.rng.standard_normal(n)).astype(np.float32)

    def uniform(self, key, shape, lo, hi):
        self.sd[key] = self.rng.uniform(lo, hi, shape).astype(np.float32)


def make_hint_state(cfg: dict | None = None, seed: int = 1, gain: float = 1.0,
                    mod_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """HINT state dict (hint_arch.py:494-655 constructors). `mod_gain` scales the
    inter-modulator kernels; > 1 drives the cache's floor-division off zero."""
    cfg = cfg or hint_default_config()
    g = _Gen(seed, gain)
    dim, bias = cfg["dim"], cfg.get("bias", False)
    with_bias_ln = cfg.get("ln_type", "WithBias") != "BiasFree"
    ffn = cfg["ffn_expansion_factor"]
    cin, cout = cfg.get("inp_channels", 3), cfg.get("out_channels", 3)

    def conv(key, co, ci, k, b):
        g.kernel(key + ".weight", (co, ci, k, k))
        if b:
            g.vec(key + ".bias", co, 0.0, 0.05)

    def ln(key, c):
        g.vec(key + ".body.weight", c, 1.0, 0.1)
        if with_bias_ln:
            g.vec(key + ".body.bias", c, 0.0, 0.1)

    def block(pre, c, is_att):
        if is_att:
            ln(pre + ".norm1", c)
            a = pre + ".attn"
            g.uniform(a + ".temperature", (4, 1, 1), 0.5, 2.0)
            conv(a + ".qkv", 3 * c, c, 1, bias)
            g.kernel(a + ".qkv_dwconv.weight", (3 * c, 1, 3, 3))
            if bias:
                g.vec(a + ".qkv_dwconv.bias", 3 * c, 0.0, 0.05)
            conv(a + ".project_out", c, c, 1, bias)
            im = a + ".intra_modulator"
            g.kernel(im + ".down.weight", (c // 2, c, 1)); g.vec(im + ".down.bias", c // 2, 0, 0.05)
            g.kernel(im + ".up.weight", (c, c // 2, 1)); g.vec(im + ".up.bias", c, 0, 0.05)
            g.kernel(im + ".gating_conv.weight", (c, c, 1)); g.vec(im + ".gating_conv.bias", c, 0, 0.05)
            for gi, r in enumerate((1, 2, 2, 3)):
                cg = r * c // 8
                m = f"{a}.inter_modulator{gi + 1}"
                g.kernel(m + ".conv_width.weight", (2 * cg, cg, 1))
                g.sd[m + ".conv_width.weight"] *= np.float32(mod_gain)
                g.vec(m + ".conv_width.bias", 2 * cg, 0, 0.05 * mod_gain)
                g.kernel(m + ".gating_conv.weight", (cg, cg, 1))
                g.sd[m + ".gating_conv.weight"] *= np.float32(mod_gain)
                g.vec(m + ".gating_conv.bias", cg, 0, 0.05 * mod_gain)
        ln(pre + ".norm2", c)
        h = int(c * ffn)
        f = pre + ".ffn"
        conv(f + ".project_in", 2 * h, c, 1, bias)
        g.kernel(f + ".dwconv.weight", (2 * h, 1, 3, 3))
        if bias:
            g.vec(f + ".dwconv.bias", 2 * h, 0.0, 0.05)
        conv(f + ".project_out", c, h, 1, bias)

    g.kernel("patch_embed.proj.weight", (dim, cin, 3, 3))
    for name, n, mult, is_att in hint_stage_plan(cfg):
        for i in range(n):
            block(f"{name}.{i}", dim * mult, is_att)
    g.kernel("down1_2.body.0.weight", (dim // 2, dim, 3, 3))
    g.kernel("down2_3.body.0.weight", (dim, 2 * dim, 3, 3))
    g.kernel("down3_4.body.0.weight", (2 * dim, 4 * dim, 3, 3))
    g.kernel("up4_3.body.0.weight", (16 * dim, 8 * dim, 3, 3))
    g.kernel("up3_2.body.0.weight", (8 * dim, 4 * dim, 3, 3))
    g.kernel("up2_1.body.0.weight", (4 * dim, 2 * dim, 3, 3))
    conv("reduce_chan_level3", 4 * dim, 8 * dim, 1, bias)
    conv("reduce_chan_level2", 2 * dim, 4 * dim, 1, bias)
    if cfg.get("dual_pixel_task", False):
        conv("skip_conv", 2 * dim, dim, 1, bias)
    conv("output", cout, 2 * dim, 3, bias)
    return g.sd


def synthetic_images(b: int, h: int, w: int, seed: int = 0, sigma: float | None = None) -> np.ndarray:
    """Uniform [0,1) NCHW fp32 images; with `sigma`, Gaussian noise sigma/255 is
    added without clamping (Denoising/test_gaussian_color_denoising_HINT.py:160-163)."""
    rng = np.random.default_rng(seed)
    x = rng.random((b, 3, h, w), dtype=np.float32)
    if sigma is not None:
        x = (x + rng.normal(0.0, sigma / 255.0, x.shape)).astype(np.float32)
    return x


def make_astv2_state(cfg: dict | None = None, seed: int = 1, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """ASTv2 state dict (astv2_arch.py:79-100,136-156,303-436 constructors)."""
    default_config, stage_plan = astv2_default_config, astv2_stage_plan
    cfg = cfg or default_config()
    g = _Gen(seed, gain)
    dim, bias = cfg["dim"], cfg.get("bias", False)
    with_bias_ln = cfg.get("ln_type", "WithBias") != "BiasFree"
    ffn = cfg["ffn_expansion_factor"]
    heads = cfg["heads"]
    cin, cout = cfg.get("inp_channels", 3), cfg.get("out_channels", 3)

    def conv(key, co, ci, k, b):
        g.kernel(key + ".weight", (co, ci, k, k))
        if b:
            g.vec(key + ".bias", co, 0.0, 0.05)

    def ln(key, c):
        g.vec(key + ".body.weight", c, 1.0, 0.1)
        if with_bias_ln:
            g.vec(key + ".body.bias", c, 0.0, 0.1)

    for name, n, mult, hi, is_att in stage_plan(cfg):
        c = dim * mult
        for i in range(n):
            pre = f"{name}.{i}"
            if is_att:
                ln(pre + ".norm1", c)
                a = pre + ".attn"
                ch = c // heads[hi]
                g.uniform(a + ".temperature", (heads[hi], 1, 1), 0.5, 2.0)
                conv(a + ".qkv", 3 * c, c, 1, bias)
                g.kernel(a + ".qkv_dwconv.weight", (3 * c, 1, 3, 3))
                if bias:
                    g.vec(a + ".qkv_dwconv.bias", 3 * c, 0.0, 0.05)
                conv(a + ".project_out", c, c, 1, bias)
                conv(a + ".att_ca", 2 * ch, ch, 1, bias)
            ln(pre + ".norm2", c)
            f = pre + ".ffn"
            h = int(c * ffn)
            g.kernel(f + ".partial_conv3.weight", (c // 24, 1, 3, 3))
            conv(f + ".project_in", 2 * h, c, 1, bias)
            g.kernel(f + ".dwconv.weight", (h, 1, 3, 3))
            if bias:
                g.vec(f + ".dwconv.bias", h, 0.0, 0.05)
            conv(f + ".project_out", c, h, 1, bias)
    g.kernel("patch_embed.proj.weight", (dim, cin, 3, 3))
    g.kernel("down1_2.body.0.weight", (dim // 2, dim, 3, 3))
    g.kernel("down2_3.body.0.weight", (dim, 2 * dim, 3, 3))
    g.kernel("up3_2.body.0.weight", (8 * dim, 4 * dim, 3, 3))
    g.kernel("up2_1.body.0.weight", (4 * dim, 2 * dim, 3, 3))
    conv("reduce_chan_level2", 2 * dim, 4 * dim, 1, bias)
    if cfg.get("dual_pixel_task", False):
        conv("skip_conv", 2 * dim, dim, 1, bias)
    conv("output", cout, 2 * dim, 3, bias)
    return g.sd


def make_ast_state(cfg: dict | None = None, seed: int = 1, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """AST state dict (ast_arch.py:95-117,219-273,453-474,561-596,600-674,940-1198 constructors),
    token MLP keys by cfg["token_mlp"] (frfn / leff / ffn).  Includes the int32 `relative_position_index`
    buffers like a saved checkpoint."""
    block_window, default_config, layer_plan = ast_block_window, ast_default_config, ast_layer_plan
    cfg = cfg or default_config()
    g = _Gen(seed, gain)
    E = cfg["embed_dim"]
    ratio = cfg.get("mlp_ratio", 4.0)
    qkv_bias = cfg.get("qkv_bias", True)

    def linear(key, co, ci, bias=True):
        g.kernel(key + ".weight", (co, ci))
        if bias:
            g.vec(key + ".bias", co, 0.0, 0.05)

    g.kernel("input_proj.proj.0.weight", (E, cfg.get("dd_in", 3), 3, 3))
    g.vec("input_proj.proj.0.bias", E, 0.0, 0.05)
    g.kernel("output_proj.proj.0.weight", (cfg.get("in_chans", 3), 2 * E, 3, 3))
    g.vec("output_proj.proj.0.bias", cfg.get("in_chans", 3), 0.0, 0.05)
    for name, mult, div, depth, heads, att in layer_plan(cfg):
        c = E * mult
        h = int(c * ratio)
        for i in range(depth):
            pre = f"{name}.blocks.{i}"
            if att:
                win, _ = block_window(cfg, div, i)
                g.vec(pre + ".norm1.weight", c, 1.0, 0.1)
                g.vec(pre + ".norm1.bias", c, 0.0, 0.1)
                a = pre + ".attn"
                g.sd[a + ".relative_position_bias_table"] = (
                    g.rng.standard_normal(((2 * win - 1) ** 2, heads)) * 0.2).astype(np.float32)
                g.sd[a + ".relative_position_index"] = ast_relative_position_index(win)
                linear(a + ".qkv.to_q", c, c, qkv_bias)
                linear(a + ".qkv.to_kv", 2 * c, c, qkv_bias)
                linear(a + ".proj", c, c)
                g.sd[a + ".w"] = g.rng.uniform(-1.0, 1.0, 2).astype(np.float32)
            g.vec(pre + ".norm2.weight", c, 1.0, 0.1)
            g.vec(pre + ".norm2.bias", c, 0.0, 0.1)
            m = pre + ".mlp"
            kind = cfg.get("token_mlp", "leff")
            if kind in ("ffn", "mlp"):
                linear(m + ".fc1", h, c)
                linear(m + ".fc2", c, h)
                continue
            linear(m + ".linear1.0", 2 * h if kind == "frfn" else h, c)
            g.kernel(m + ".dwconv.0.weight", (h, 1, 3, 3))
            g.vec(m + ".dwconv.0.bias", h, 0.0, 0.05)
            linear(m + ".linear2.0", c, h)
            if kind == "frfn":
                g.kernel(m + ".partial_conv3.weight", (c // 4, c // 4, 3, 3))
    for i, (ci, co) in enumerate(((E, 2 * E), (2 * E, 4 * E), (4 * E, 8 * E), (8 * E, 16 * E))):
        g.kernel(f"dowsample_{i}.conv.0.weight", (co, ci, 4, 4))
        g.vec(f"dowsample_{i}.conv.0.bias", co, 0.0, 0.05)
    for i, (ci, co) in enumerate(((16 * E, 8 * E), (16 * E, 4 * E), (8 * E, 2 * E), (4 * E, E))):
        g.kernel(f"upsample_{i}.deconv.0.weight", (ci, co, 2, 2), fan_in=ci)
        g.vec(f"upsample_{i}.deconv.0.bias", co, 0.0, 0.05)
    return g.sd


def make_fpro_state(cfg: dict | None = None, seed: int = 1, gain: float = 1.0,
                    prompt_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """FPro state dict (fpro_arch.py:92-161,212-221,249-308,339-354,378-400,452-469,548-682
    constructors).  The learned prompt maps (`FSPG_high.weight`, `FSPG_low.complex_weight`, 0.02 *
    randn in the reference) are drawn with std `prompt_gain` so the prompt branches carry weight in
    a parity test; BatchNorm running statistics are randomised like a trained checkpoint's."""
    FILTER_GROUPS, default_config, prompt_plan, stage_plan = (FPRO_FILTER_GROUPS, fpro_default_config, fpro_prompt_plan,
                                                              fpro_stage_plan)
    cfg = cfg or default_config()
    g = _Gen(seed, gain)
    dim, bias = cfg["dim"], cfg.get("bias", False)
    with_bias_ln = cfg.get("ln_type", "WithBias") != "BiasFree"
    ffn = cfg["ffn_expansion_factor"]
    heads = cfg["heads"]
    cin, cout = cfg.get("inp_channels", 3), cfg.get("out_channels", 3)

    def conv(key, co, ci, k, b):
        g.kernel(key + ".weight", (co, ci, k, k))
        if b:
            g.vec(key + ".bias", co, 0.0, 0.05)

    def linear(key, co, ci):
        g.kernel(key + ".weight", (co, ci))
        g.vec(key + ".bias", co, 0.0, 0.05)

    def ln(key, c):
        g.vec(key + ".body.weight", c, 1.0, 0.1)
        if with_bias_ln:
            g.vec(key + ".body.bias", c, 0.0, 0.1)

    for name, n, mult, hi, is_att in stage_plan(cfg):
        c = dim * mult
        for i in range(n):
            pre = f"{name}.{i}"
            if is_att:
                ln(pre + ".norm1", c)
                a = pre + ".attn"
                g.uniform(a + ".temperature", (heads[hi], 1, 1), 0.5, 2.0)
                conv(a + ".qkv", 3 * c, c, 1, bias)
                g.kernel(a + ".qkv_dwconv.weight", (3 * c, 1, 3, 3))
                if bias:
                    g.vec(a + ".qkv_dwconv.bias", 3 * c, 0.0, 0.05)
                conv(a + ".project_out", c, c, 1, bias)
            ln(pre + ".norm2", c)
            f = pre + ".ffn"
            h = int(c * ffn)
            conv(f + ".project_in", 2 * h, c, 1, bias)
            g.kernel(f + ".dwconv.weight", (2 * h, 1, 3, 3))
            if bias:
                g.vec(f + ".dwconv.bias", 2 * h, 0.0, 0.05)
            conv(f + ".project_out", c, h, 1, bias)
    g.kernel("patch_embed.proj.weight", (dim, cin, 3, 3))
    g.kernel("down1_2.body.0.weight", (dim // 2, dim, 3, 3))
    g.kernel("down2_3.body.0.weight", (dim, 2 * dim, 3, 3))
    g.kernel("up3_2.body.0.weight", (8 * dim, 4 * dim, 3, 3))
    g.kernel("up2_1.body.0.weight", (4 * dim, 2 * dim, 3, 3))
    conv("reduce_chan_level2", 2 * dim, 4 * dim, 1, bias)
    if cfg.get("dual_pixel_task", False):
        conv("skip_conv", 2 * dim, dim, 1, bias)
    conv("output", cout, 2 * dim, 3, bias)

    d = "splitFre.dyna_channel"
    nf = FILTER_GROUPS * 9
    g.kernel(d + ".conv.weight", (nf, dim, 1, 1))
    # the pooled feature is small: scale the filter logits up so the 9 taps are visibly unequal
    g.sd[d + ".conv.weight"] *= np.float32(8.0)
    g.kernel(d + ".conv_gate.weight", (nf, nf, 1, 1))
    g.vec(d + ".bn.weight", nf, 1.0, 0.2)
    g.vec(d + ".bn.bias", nf, 0.0, 0.2)
    g.vec(d + ".bn.running_mean", nf, 0.0, 0.2)
    g.uniform(d + ".bn.running_var", (nf,), 0.5, 1.5)

    for name, mult, res in prompt_plan(cfg):
        c = dim * mult
        conv(name + ".simple_Fusion", c, 2 * c, 1, True)
        hi_ = name + ".FSPG_high"
        # (x 30: the high-frequency prompt is a product of three small factors)
        g.sd[hi_ + ".weight"] = (30.0 * prompt_gain * g.rng.standard_normal((1, dim, res, res))).astype(np.float32)
        g.kernel(hi_ + ".body.0.weight", (dim, 1, 1, 3))
        g.vec(hi_ + ".body.0.bias", dim, 0.0, 0.05)
        g.kernel(hi_ + ".body.1.weight", (dim, 1, 3, 1))
        g.vec(hi_ + ".body.1.bias", dim, 0.0, 0.05)
        lo_ = name + ".FSPG_low"
        g.sd[lo_ + ".complex_weight"] = (prompt_gain * g.rng.standard_normal((1, dim, res, res // 2 + 1, 2))).astype(np.float32)
        g.kernel(lo_ + ".body.0.weight", (2 * dim, 2 * dim, 1, 1))
        # the spectrum's DC term is a sum over all pixels: keep the gate's argument O(1)
        g.sd[lo_ + ".body.0.weight"] *= np.float32(1e-3)
        g.vec(lo_ + ".body.0.bias", 2 * dim, 0.0, 0.05)
        mh = name + ".modulator_hi"
        linear(mh + ".to_q.to_q", c, c)
        linear(mh + ".to_kv.to_kv", 2 * c, dim)
        g.sd[mh + ".to_kv.to_kv.weight"] *= np.float32(8.0)     # the prompt is small: let k, v depend on it
        g.sd[mh + ".to_q.to_q.weight"] *= np.float32(3.0)       # and the softmax on q
        g.kernel(mh + ".kv_dwconv.weight", (dim, 1, 3, 3))
        linear(mh + ".project_out", c, c)
        ml = name + ".modulator_lo"
        g.uniform(ml + ".temperature", (2, 1, 1), 0.5, 2.0)
        conv(ml + ".q", c, c, 1, False)
        conv(ml + ".kv", 2 * c, dim, 1, False)
        conv(ml + ".project_out", c, c, 1, False)
    return g.sd


def default_config(arch: str) -> dict:
    """The `network_g` options the reference's YAMLs use for `arch`."""
    return {"HINT": hint_default_config, "ASTv2": astv2_default_config, "AST": ast_default_config,
            "FPro": fpro_default_config}[arch]()


def make_state(arch: str, cfg: dict | None = None, seed: int = 1, gain: float | None = None):
    """Seeded state dict for `arch` (gain defaults to the value the parity tests use)."""
    if arch == "HINT":
        return make_hint_state(cfg, seed=seed, gain=0.6 if gain is None else gain)
    if arch == "ASTv2":
        return make_astv2_state(cfg, seed=seed, gain=0.6 if gain is None else gain)
    if arch == "AST":
        return make_ast_state(cfg, seed=seed, gain=0.5 if gain is None else gain)
    if arch == "FPro":
        return make_fpro_state(cfg, seed=seed, gain=0.6 if gain is None else gain)
    raise ValueError(f"{arch} is not found.")
