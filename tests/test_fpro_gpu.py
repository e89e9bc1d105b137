"""FPro parity on the GPU through the drop-in class, against the oracle (max-abs <= 1e-3).

The oracle runs the reference's literal computation, including the full FFT round trip of the
low-frequency prompt; the HIP path uses the reduction described in fpro_ops.hip (only the spectrum's
DC term reaches the network), so agreement here also checks that reduction."""
import numpy as np
import pytest
import torch

from oracle import fpro_oracle as fo
from oracle.common import psnr
from oracle.weights import make_fpro_state, synthetic_images
from tests.gpu_util import max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-3

SMALL = dict(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 2, 2], num_refinement_blocks=1,
             heads=[1, 2, 4], ffn_expansion_factor=2.66, bias=False, ln_type="WithBias", dual_pixel_task=False)


def _build(cfg, sd):
    from basicsr.models.archs import define_network
    net = define_network(dict(cfg, type="FPro"))
    net.load_state_dict(sd)
    return net.cuda().eval()


@pytest.mark.parametrize("name,cfg_update,B,H,W", [
    ("small_64x96", {}, 2, 64, 96),
    ("small_bias_biasfree_ln", dict(bias=True, ln_type="BiasFree"), 1, 96, 64),
    ("small_dual_pixel", dict(dual_pixel_task=True, inp_channels=6), 1, 64, 64),
    ("default_256", None, 1, 256, 256),        # the tile size every FPro script uses
    ("default_128x160", None, 1, 128, 160),    # prompt maps resized to a non-nominal resolution
])
def test_network(name, cfg_update, B, H, W):
    cfg = fo.default_config() if cfg_update is None else dict(SMALL, **cfg_update)
    sd = make_fpro_state(cfg, seed=11, gain=0.6)
    x = synthetic_images(B, H, W, seed=51, sigma=25)
    if cfg["inp_channels"] != 3:
        x = np.concatenate([x, synthetic_images(B, H, W, seed=52)], axis=1)
    y_ref = fo.fpro_forward(sd, torch.from_numpy(x), cfg).numpy()
    y = _build(cfg, sd)(torch.from_numpy(x).cuda()).cpu().numpy()
    err = max_abs(y, y_ref)
    print(f"[{name}] max|dy| = {err:.2e}, |y| max {np.abs(y_ref).max():.2f}")
    assert err < TOL
    if cfg["inp_channels"] == 3:
        assert abs(psnr(y, x) - psnr(y_ref, x)) < 0.01


def test_prompts_matter_and_low_branch_reduction_holds():
    """Zeroing the prompt maps changes the output (the prompt path is live in the test weights), and
    the oracle's own low-frequency prompt has the spatial mean the HIP path derives from the DC term."""
    cfg = dict(SMALL)
    sd = make_fpro_state(cfg, seed=12, gain=0.6)
    x = synthetic_images(1, 64, 64, seed=53, sigma=25)
    taps = {}
    y_ref = fo.fpro_forward(sd, torch.from_numpy(x), cfg, tap=lambda n, t: taps.__setitem__(n, t)).numpy()
    sd0 = dict(sd)
    for k in sd:
        if k.endswith("FSPG_high.weight") or k.endswith("FSPG_low.complex_weight"):
            sd0[k] = np.zeros_like(sd[k])
    y0 = fo.fpro_forward(sd0, torch.from_numpy(x), cfg).numpy()
    assert max_abs(y0, y_ref) > 1e-2
    assert float(taps["prompt_r.lo"].std(dim=(2, 3)).max()) < 1e-5     # spatially constant
    y = _build(cfg, sd)(torch.from_numpy(x).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL


def test_rejects_bad_shapes():
    net = _build(SMALL, make_fpro_state(SMALL, seed=1))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 48, 64).cuda())


def test_sots_sized_image_through_the_tiled_script_path():
    """A 620 x 460 image the way Dehaze/test_SOTS.py runs FPro: pad to 624 x 464, 20 tiles of 256 x 256
    with overlap 158, one forward per tile, mean blend, unpad, clamp (test_rain100L.py:156-175 has the
    same branch).  The device composite equals the same steps done with the oracle's plumbing around
    the same network."""
    from jirf_hip import harness as hh
    from oracle import harness_oracle as ho
    cfg = dict(SMALL)
    net = _build(cfg, make_fpro_state(cfg, seed=13, gain=0.6))
    net.batch_mode = "per_image"
    x = synthetic_images(1, 460, 620, seed=54, sigma=10)
    got = hh.restore_image(net, torch.from_numpy(x).cuda(), crop_size=256, overlap_size=158, tile_batch=8).cpu().numpy()
    xp = ho.pad_to_factor(x)
    split, starts = ho.splitimage(xp, 256, 158)
    assert xp.shape[2:] == (464, 624) and len(split) == 20
    outs = [net(torch.from_numpy(np.ascontiguousarray(t)).cuda()).cpu().numpy() for t in split]
    want = ho.crop_clamp(ho.mergeimage(outs, starts, 256, xp.shape), 460, 620)
    # tiles run 8 at a time here and one at a time there: the split-K schedule of the channel
    # attention depends on the batch, so sums differ in their last bits
    assert max_abs(got, want) < 1e-5


def test_default_fpro_sots_sized_image_against_twenty_oracle_tile_calls():
    """BASELINE configs[4], FPro half, end to end: the reference-default FPro (dim 48, blocks [2,3,6], ffn 3) on a
    620 x 460 image through `restore_image` (pad to 624 x 464, 20 tiles of 256 x 256 overlapping by 158, mean blend,
    unpad, clamp) against the ORACLE network called once per tile - what Dehaze/test_SOTS.py:132-180 does with the
    reference network - with the oracle's own NumPy plumbing around it."""
    from jirf_hip import harness as hh
    from oracle import harness_oracle as ho
    cfg = fo.default_config()
    sd = make_fpro_state(cfg, seed=14, gain=0.6)
    net = _build(cfg, sd)
    x = synthetic_images(1, 460, 620, seed=55, sigma=10)
    got = hh.restore_image(net, torch.from_numpy(x).cuda(), crop_size=256, overlap_size=158).cpu().numpy()
    xp = ho.pad_to_factor(x)
    split, starts = ho.splitimage(xp, 256, 158)
    assert xp.shape[2:] == (464, 624) and len(split) == 20
    outs = [fo.fpro_forward(sd, torch.from_numpy(np.ascontiguousarray(t)), cfg).numpy() for t in split]
    want = ho.crop_clamp(ho.mergeimage(outs, starts, 256, xp.shape), 460, 620)
    err = max_abs(got, want)
    print(f"[FPro default, 620x460 as 20 tiles] max|dy| = {err:.2e} vs 20 oracle tile calls")
    assert err < TOL
    assert abs(psnr(got, x) - psnr(want, x)) < 0.01


@pytest.mark.parametrize("B,C,heads,H,W", [
    (1, 96, 2, 64, 64),      # the prompt modules after decoder_level2 / 1 / refinement: 48 channels per head (matrix-core kernel)
    (2, 192, 2, 16, 24),     # after decoder_level3: 96 channels per head (the vector kernel)
    (1, 64, 2, 8, 40),       # 32 channels per head, one row of windows
])
def test_cross_window_attention(B, C, heads, H, W):
    """highFrequencyPromptFusion's 8x8-window cross attention at kernel level, including the reference's
    transpose-then-reshape flatten (fpro_arch.py:322-332)."""
    from tests.gpu_util import call, dev, ptr, stream
    rng = np.random.default_rng(C + H)
    q = rng.standard_normal((B, C, H, W)).astype(np.float32)
    kv = rng.standard_normal((B, 2 * C, H, W)).astype(np.float32)
    qt = fo.window_partition(torch.from_numpy(q).permute(0, 2, 3, 1), fo.WIN).reshape(-1, 64, C)
    kt = fo.window_partition(torch.from_numpy(kv).permute(0, 2, 3, 1), fo.WIN).reshape(-1, 64, 2 * C)
    ref = fo.cross_window_core(qt, kt, heads)
    ref = fo.window_reverse(ref.reshape(-1, fo.WIN, fo.WIN, C), fo.WIN, H, W).permute(0, 3, 1, 2).numpy()
    out = torch.empty(B, C, H, W, device="cuda")
    qd, kd = dev(q), dev(kv)
    call("jirf_op_cross_window_attention", ptr(qd), ptr(kd), ptr(out), B, C, heads, H, W, stream())
    assert max_abs(out.cpu().numpy(), ref) < 2e-5 * max(1.0, float(np.abs(ref).max()))
    # the flatten is not the plain one: a kernel that permuted the heads back would differ visibly
    hd = C // heads
    qh = qt.reshape(-1, 64, heads, hd).permute(0, 2, 1, 3) * hd ** -0.5
    kh = kt[..., :C].reshape(-1, 64, heads, hd).permute(0, 2, 1, 3)
    vh = kt[..., C:].reshape(-1, 64, heads, hd).permute(0, 2, 1, 3)
    plain = ((qh @ kh.transpose(-2, -1)).softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(-1, 64, C)
    plain = fo.window_reverse(plain.reshape(-1, fo.WIN, fo.WIN, C), fo.WIN, H, W).permute(0, 3, 1, 2).numpy()
    assert max_abs(plain, ref) > 0.1


@pytest.mark.parametrize("B,C,H,W", [(1, 48, 64, 64), (2, 48, 24, 40), (1, 16, 8, 12)])
def test_dynamic_filter(B, C, H, W):
    """dynamic_filter_channel (fpro_arch.py:355-375) at kernel level: the per-image 3x3 low-pass whose taps come
    from the pooled image through conv / gate / BatchNorm / softmax, applied with reflection padding."""
    from tests.gpu_util import call, dev, hptr, ptr, stream
    rng = np.random.default_rng(C + H + W)
    g = fo.FILTER_GROUPS
    F9 = 9 * g
    x = rng.standard_normal((B, C, H, W)).astype(np.float32) + rng.standard_normal((B, C, 1, 1)).astype(np.float32)
    prm = {"d.conv.weight": rng.standard_normal((F9, C, 1, 1)).astype(np.float32) * 0.5,
           "d.conv_gate.weight": rng.standard_normal((F9, F9, 1, 1)).astype(np.float32) * 0.3,
           "d.bn.weight": (1.0 + 0.3 * rng.standard_normal(F9)).astype(np.float32),
           "d.bn.bias": (0.2 * rng.standard_normal(F9)).astype(np.float32),
           "d.bn.running_mean": (0.1 * rng.standard_normal(F9)).astype(np.float32),
           "d.bn.running_var": (0.5 + rng.random(F9)).astype(np.float32)}
    low_ref, high_ref = fo.dynamic_filter({k: torch.from_numpy(v) for k, v in prm.items()}, "d", torch.from_numpy(x))
    flat = np.concatenate([prm[k].ravel() for k in ("d.conv.weight", "d.conv_gate.weight", "d.bn.weight", "d.bn.bias",
                                                    "d.bn.running_mean", "d.bn.running_var")]).astype(np.float32)
    low = torch.empty(B, C, H, W, device="cuda")
    high = torch.empty(B, C, H, W, device="cuda")
    xd = dev(x)
    call("jirf_op_dynamic_filter", ptr(xd), hptr(flat), g, ptr(low), ptr(high), B, C, H, W, stream())
    assert max_abs(low.cpu().numpy(), low_ref.numpy()) < 1e-5 * max(1.0, float(low_ref.abs().max()))
    assert max_abs(high.cpu().numpy(), high_ref.numpy()) < 1e-5 * max(1.0, float(high_ref.abs().max()))
    # the filter differs from image to image and is not the identity
    assert float(high_ref.abs().mean()) > 0.05


@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(64, 64, 32, 32), (64, 96, 16, 24), (128, 128, 64, 64), (32, 32, 64, 64),
                                         (24, 40, 36, 52), (256, 256, 128, 128)])
@pytest.mark.parametrize("bicubic", [0, 1])
def test_resize(Hi, Wi, Ho, Wo, bicubic):
    """The two interpolations of the prompt generator (fpro_arch.py:405,414,425-436) against torch's
    F.interpolate(align_corners=False): bilinear for the high part and the learned maps, bicubic for the low part.
    (That Jittor's bicubic equals torch's is one of the named, unverifiable assumptions of the oracle.)"""
    import torch.nn.functional as F
    from tests.gpu_util import call, dev, ptr, stream
    rng = np.random.default_rng(Hi + Wo + bicubic)
    x = rng.standard_normal((2, 5, Hi, Wi)).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(x), size=(Ho, Wo), mode="bicubic" if bicubic else "bilinear", align_corners=False)
    y = torch.empty(2, 5, Ho, Wo, device="cuda")
    xd = dev(x)
    call("jirf_op_resize", ptr(xd), ptr(y), 2, 5, Hi, Wi, Ho, Wo, bicubic, stream())
    assert max_abs(y.cpu().numpy(), ref.numpy()) < 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,C,H,W", [(1, 48, 64, 64), (2, 48, 16, 24), (1, 8, 8, 8)])
def test_high_frequency_prompt(B, C, H, W):
    """frequenctSpecificPromptGenetator's high branch at kernel level (fpro_arch.py:404-411): depthwise 1x3, depthwise
    3x1, GELU, times the input and the learned map (both given at the level's size, so the oracle's resizes are the
    identity)."""
    from tests.gpu_util import call, dev, ptr, stream
    rng = np.random.default_rng(C + H)
    f = rng.standard_normal((B, C, H, W)).astype(np.float32)
    prm = {"g.body.0.weight": rng.standard_normal((C, 1, 1, 3)).astype(np.float32) * 0.5,
           "g.body.0.bias": (0.1 * rng.standard_normal(C)).astype(np.float32),
           "g.body.1.weight": rng.standard_normal((C, 1, 3, 1)).astype(np.float32) * 0.5,
           "g.body.1.bias": (0.1 * rng.standard_normal(C)).astype(np.float32),
           "g.weight": rng.standard_normal((1, C, H, W)).astype(np.float32)}
    ref = fo.fspg_high({k: torch.from_numpy(v) for k, v in prm.items()}, "g", torch.from_numpy(f), H, W)
    y = torch.empty(B, C, H, W, device="cuda")
    d = {k: dev(v) for k, v in prm.items()}
    fd = dev(f)
    call("jirf_op_fspg_high", ptr(fd), ptr(d["g.body.0.weight"]), ptr(d["g.body.0.bias"]), ptr(d["g.body.1.weight"]),
         ptr(d["g.body.1.bias"]), ptr(d["g.weight"]), ptr(y), B, C, H, W, stream())
    assert max_abs(y.cpu().numpy(), ref.numpy()) < 1e-5 * max(1.0, float(ref.abs().max()))
