"""The HIP path against the committed fixtures (tests/golden/*.npz, written by tests/make_golden.py
from the oracle): fixed inputs, fixed expected outputs, no oracle code in the loop."""
import os

import numpy as np
import pytest
import torch

from oracle.weights import make_ast_state, make_astv2_state, make_fpro_state, make_hint_state
from tests.gpu_util import max_abs
from tests.make_golden import AST_SHALLOW, ASTV2_SMALL, FPRO_SMALL, HINT_SMALL, weights_digest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3


def _net(arch, cfg, sd, batch_mode="coupled"):
    from basicsr.models.archs import define_network
    net = define_network(dict(cfg, type=arch))
    net.load_state_dict(sd)
    net.batch_mode = batch_mode
    return net.cuda().eval()


def _split(flat, sizes):
    out, o = [], 0
    for n in sizes:
        out.append(flat[o:o + n][None])
        o += n
    return out


@pytest.mark.parametrize("name,cfg", [("hint_small_48x64", HINT_SMALL), ("hint_full_32x32", None),
                                      ("hint_small_modgain_32x48", HINT_SMALL)])
def test_hint_golden(name, cfg):
    from oracle.hint_oracle import default_config
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = cfg or default_config()
    sd = make_hint_state(cfg, seed=int(g["seed"]), gain=0.6, mod_gain=float(g["mod_gain"]))
    assert weights_digest(sd) == str(g["weights_sha256"])
    net = _net("HINT", cfg, sd)
    _, _, H, W = g["x1"].shape
    net.native().plan(1, H, W, 1)
    # cold call, then the warm call that continues from the fixture's cache (orders pinned: the
    # argsort behind them is discontinuous, see the attention tests)
    net.force_perms(_split(g["perms1"], g["perm_sizes"]))
    y1 = net(torch.from_numpy(g["x1"]).cuda()).cpu().numpy()
    assert max_abs(y1, g["y1"]) < TOL
    assert float(np.mean(np.abs(net.qv_cache - g["cache1"]) > 1e-5)) < 1e-3
    net.qv_cache = g["cache1"]
    net.force_perms(_split(g["perms2"], g["perm_sizes"]))
    y2 = net(torch.from_numpy(g["x2"]).cuda()).cpu().numpy()
    assert max_abs(y2, g["y2"]) < TOL
    assert float(np.mean(np.abs(net.qv_cache - g["cache2"]) > 1e-5)) < 1e-3


def test_astv2_golden():
    g = np.load(os.path.join(GOLD, "astv2_small_40x56.npz"))
    sd = make_astv2_state(ASTV2_SMALL, seed=int(g["seed"]), gain=0.6)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = _net("ASTv2", ASTV2_SMALL, sd)(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert max_abs(y, g["y"]) < TOL


def test_ast_golden():
    g = np.load(os.path.join(GOLD, "ast_shallow_128.npz"))
    sd = make_ast_state(AST_SHALLOW, seed=int(g["seed"]), gain=0.5)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = _net("AST", AST_SHALLOW, sd)(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert max_abs(y, g["y"]) < TOL


def test_fpro_golden():
    g = np.load(os.path.join(GOLD, "fpro_small_64x96.npz"))
    sd = make_fpro_state(FPRO_SMALL, seed=int(g["seed"]), gain=0.6)
    assert weights_digest(sd) == str(g["weights_sha256"])
    y = _net("FPro", FPRO_SMALL, sd)(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert max_abs(y, g["y"]) < TOL
