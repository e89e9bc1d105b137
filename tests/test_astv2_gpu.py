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
