"""AST parity on the GPU through the drop-in class, against the oracle (max-abs <= 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import ast_oracle as so
from oracle.common import psnr
from oracle.weights import make_ast_state, synthetic_images
from tests.gpu_util import max_abs

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _build(cfg, sd, batch_mode="coupled"):
    from basicsr.models.archs import define_network
    net = define_network(dict(cfg, type="AST"))
    net.load_state_dict(sd)
    net.batch_mode = batch_mode
    return net.cuda().eval()


@pytest.mark.parametrize("name,cfg_update,S", [
    ("default_128", {}, 128),
    ("default_256", {}, 256),                                   # BASELINE config 0: AST-B on one 256x256 image
    ("nominal_128", dict(img_size=128), 128),                   # bottleneck nominal resolution 8: never shifts
    ("shallow_no_shift", dict(depths=[1] * 9, shift_flag=False), 128),
    ("deep_bottleneck", dict(depths=[1, 1, 1, 1, 4, 2, 2, 2, 2]), 128),
])
def test_network_b1(name, cfg_update, S):
    cfg = dict(so.default_config(), **cfg_update)
    sd = make_ast_state(cfg, seed=5, gain=0.5)
    x = synthetic_images(1, S, S, seed=41)
    y_ref = so.ast_forward(sd, torch.from_numpy(x), cfg).numpy()
    y = _build(cfg, sd)(torch.from_numpy(x).cuda()).cpu().numpy()
    err = max_abs(y, y_ref)
    print(f"[{name}] max|dy| = {err:.2e}, |y| max {np.abs(y_ref).max():.2f}")
    assert err < TOL
    assert abs(psnr(y, x) - psnr(y_ref, x)) < 0.01


@pytest.mark.parametrize("token_mlp", ["leff", "ffn", "mlp"])
def test_other_token_mlps(token_mlp):
    """The token MLPs the shipped YAML does not select (ast_arch.py:741-757): 'leff' is what the bare
    constructor AST() builds (LeFF, :407-450), 'ffn' / 'mlp' the plain Mlp (:378-404)."""
    from basicsr.models.archs import define_network
    cfg = dict(so.default_config(), token_mlp=token_mlp)
    sd = make_ast_state(cfg, seed=7, gain=0.5)
    x = synthetic_images(1, 128, 128, seed=44)
    y_ref = so.ast_forward(sd, torch.from_numpy(x), cfg).numpy()
    net = define_network(dict(type="AST")) if token_mlp == "leff" else define_network(dict(cfg, type="AST"))
    net.load_state_dict(sd)
    y = net.cuda().eval()(torch.from_numpy(x).cuda()).cpu().numpy()
    err = max_abs(y, y_ref)
    print(f"[token_mlp={token_mlp}] max|dy| = {err:.2e}")
    assert err < TOL


@pytest.mark.parametrize("batch_mode", ["coupled", "per_image"])
def test_batch_semantics(batch_mode):
    """bmm_4d pairs q with k of another (window, head) over the whole call: a literal B=2 call mixes
    images ("coupled"); "per_image" equals two B=1 calls."""
    cfg = so.default_config()
    sd = make_ast_state(cfg, seed=6, gain=0.5)
    x = synthetic_images(2, 128, 128, seed=42)
    if batch_mode == "coupled":
        y_ref = so.ast_forward(sd, torch.from_numpy(x), cfg).numpy()
    else:
        y_ref = np.concatenate([so.ast_forward(sd, torch.from_numpy(x[i:i + 1]), cfg).numpy() for i in range(2)])
    y = _build(cfg, sd, batch_mode)(torch.from_numpy(x).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL
    other = so.ast_forward(sd, torch.from_numpy(x[:1]), cfg).numpy()
    if batch_mode == "coupled":
        assert max_abs(y[:1], other) > 1e-4      # the coupling is real


def test_rejects_what_the_reference_cannot_run():
    cfg = so.default_config()
    net = _build(cfg, make_ast_state(cfg, seed=1))
    for shape in ((1, 3, 128, 256), (1, 3, 64, 64), (1, 3, 200, 200)):
        with pytest.raises(ValueError):
            net(torch.zeros(*shape).cuda())
    from basicsr.models.archs import define_network
    with pytest.raises(Exception, match="FFN error"):
        define_network(dict(type="AST", token_mlp="nope"))
