"""ASTv2 parity on the GPU through the drop-in class, against the oracle (max-abs <= 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import astv2_oracle as ao
from oracle.common import psnr
from oracle.weights import make_astv2_state, synthetic_images
from tests.gpu_util import max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-3

SMALL = dict(inp_channels=3, out_channels=3, dim=24, num_blocks=[1, 2, 2], num_refinement_blocks=1,
             heads=[1, 2, 4], ffn_expansion_factor=2.5, bias=False, ln_type="WithBias", dual_pixel_task=False)


def _build(cfg, sd):
    from basicsr.models.archs import define_network
    net = define_network(dict(cfg, type="ASTv2"))
    net.load_state_dict(sd)
    return net.cuda().eval()


@pytest.mark.parametrize("name,cfg_update,B,H,W", [
    ("small", {}, 1, 64, 96),
    ("small_ragged", {}, 2, 36, 52),                                  # H, W multiples of 4 only
    ("small_bias_biasfree", dict(bias=True, ln_type="BiasFree"), 1, 40, 48),
    ("small_dual_pixel", dict(dual_pixel_task=True, inp_channels=6), 1, 32, 32),
    ("full", None, 1, 64, 88),                                        # shipped YAML configuration
    ("full_b2", None, 2, 48, 48),
])
def test_network(name, cfg_update, B, H, W):
    cfg = ao.default_config() if cfg_update is None else dict(SMALL, **cfg_update)
    sd = make_astv2_state(cfg, seed=7, gain=0.6)
    x = synthetic_images(B, H, W, seed=31)
    if cfg["inp_channels"] == 6:
        x = np.concatenate([x, synthetic_images(B, H, W, seed=32)], axis=1)
    y_ref = ao.astv2_forward(sd, torch.from_numpy(x), cfg).numpy()
    net = _build(cfg, sd)
    y = net(torch.from_numpy(x).cuda()).cpu().numpy()
    err = max_abs(y, y_ref)
    print(f"[{name}] max|dy| = {err:.2e}, |y| max {np.abs(y_ref).max():.2f}")
    assert err < TOL
    if cfg["inp_channels"] == 3:
        assert abs(psnr(y, x) - psnr(y_ref, x)) < 0.01
    # same call again: no hidden state
    y2 = net(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(y, y2)


def test_rain100l_sized_image():
    """481x321 padded to 488x328 by the scripts' factor-8 rule (Deraining/test_rain100L.py:148-152)."""
    cfg = ao.default_config()
    sd = make_astv2_state(cfg, seed=8, gain=0.6)
    x = synthetic_images(1, 328, 488, seed=33)
    y_ref = ao.astv2_forward(sd, torch.from_numpy(x), cfg).numpy()
    y = _build(cfg, sd)(torch.from_numpy(x).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL
    assert abs(psnr(y, x) - psnr(y_ref, x)) < 0.01


def test_rejects_bad_shapes():
    net = _build(SMALL, make_astv2_state(SMALL, seed=1))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 30, 32).cuda())


@pytest.mark.parametrize("B,C,heads,P,bias,mdta", [
    (1, 96, 2, 64 * 64, False, False),     # decoder_level1 / refinement of the shipped config: 48 channels per head
    (2, 192, 8, 24 * 40, True, False),     # decoder_level3: 24 channels per head, with the att_ca bias
    (1, 96, 4, 1000, False, False),        # decoder_level2; pixel count not a multiple of the staging chunk
    (2, 96, 2, 48 * 48, False, True),      # FPro's MDTA: softmax only, x / (|x| + 1e-12) rows
    (1, 192, 4, 32 * 32, False, True),
])
def test_head_attention_core(B, C, heads, P, bias, mdta):
    """The head-wise channel attention between its two convolutions, at kernel level: statistics (split-K Q K^T,
    row norms), the per-head kernel (softmax branch, relu^2 -> gelu branch through att_ca over the ROW axis,
    scale / shift), attention * V.  ASTv2: astv2_arch.py:170-202; FPro (mdta): fpro_arch.py:134-142."""
    from oracle import fpro_oracle as fo
    from tests.gpu_util import call, dev, hptr, ptr, stream
    rng = np.random.default_rng(C + heads + P)
    ch = C // heads
    # q and k rows share a few components, so that the normalised products spread over (-1, 1) and the attention
    # is far from uniform
    comp = rng.standard_normal((B, 4, P)).astype(np.float32)
    load = rng.standard_normal((2 * C, 4)).astype(np.float32)
    qkv = (0.3 * rng.standard_normal((B, 3 * C, P))).astype(np.float32)
    qkv[:, :2 * C] += np.einsum("cr,brp->bcp", load, comp).astype(np.float32)
    temp = (1.0 + 0.5 * rng.standard_normal(heads)).astype(np.float32) * 4.0
    ca_w = (rng.standard_normal((2 * ch, ch)) / np.sqrt(ch)).astype(np.float32)
    ca_b = (0.1 * rng.standard_normal(2 * ch)).astype(np.float32) if bias else None
    q, k, v = [torch.from_numpy(qkv[:, i * C:(i + 1) * C]) for i in range(3)]
    if mdta:
        att_ref, out_ref = fo.attention_core(q, k, v, torch.from_numpy(temp), heads)
    else:
        att_ref, out_ref = ao.attention_core(q, k, v, torch.from_numpy(temp), torch.from_numpy(ca_w).reshape(2 * ch, ch, 1, 1),
                                             None if ca_b is None else torch.from_numpy(ca_b), heads)
    out = torch.empty(B, C, P, device="cuda")
    att = torch.full((B, C, C), 7.0, device="cuda")
    qd = dev(qkv)
    call("jirf_op_head_attention_core", ptr(qd), hptr(temp), None if mdta else hptr(ca_w),
         None if (mdta or ca_b is None) else hptr(ca_b), 1e-12 if mdta else 0.0, ptr(out), ptr(att), B, C, P, heads, stream())
    a = att.cpu().numpy()
    mag = max(1.0, float(att_ref.abs().max()))
    for h in range(heads):
        blk = a[:, h * ch:(h + 1) * ch, h * ch:(h + 1) * ch]
        assert max_abs(blk, att_ref[:, h].numpy()) < 1e-5 * mag
        a[:, h * ch:(h + 1) * ch, h * ch:(h + 1) * ch] = 0.0
    assert float(np.abs(a).max()) == 0.0, "attention outside the head blocks must be zero"
    assert max_abs(out.cpu().numpy(), out_ref.numpy()) < 2e-5 * max(1.0, float(out_ref.abs().max()))
    assert float(att_ref.std()) > 0.2 / ch, "test did not drive the attention off uniform"
