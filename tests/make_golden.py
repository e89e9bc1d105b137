#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (run from the repo root: python tests/make_golden.py).

The reference ships no golden vectors and its Jittor runtime is not installable here, so these
fixtures pin the ORACLE (against accidental edits) and give the GPU tests fixed expected outputs;
they are not outputs of the reference itself ("parity unpinned").  Weights are regenerated from
their seed by oracle.weights (a checksum of them is stored in the fixture).
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ast_oracle, astv2_oracle, fpro_oracle  # noqa: E402
from oracle import hint_oracle as ho  # noqa: E402
from oracle.weights import (make_ast_state, make_astv2_state, make_fpro_state, make_hint_state,  # noqa: E402
                            synthetic_images)

OUT = os.path.join(ROOT, "tests", "golden")

HINT_SMALL = dict(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 2, 2, 3], num_refinement_blocks=1,
                  heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, ln_type="WithBias",
                  dual_pixel_task=False)


ASTV2_SMALL = dict(inp_channels=3, out_channels=3, dim=24, num_blocks=[1, 2, 2], num_refinement_blocks=1,
                   heads=[1, 2, 4], ffn_expansion_factor=2.5, bias=False, ln_type="WithBias", dual_pixel_task=False)
# AST: the reference default constructor with one block per stage except two at the first (so a
# shifted block exists); 128 x 128 is the smallest input its four scales of 8 x 8 windows take
AST_SHALLOW = dict(ast_oracle.default_config(), depths=[2, 1, 1, 1, 1, 1, 1, 1, 1])


FPRO_SMALL = dict(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 2, 2], num_refinement_blocks=1,
                  heads=[1, 2, 4], ffn_expansion_factor=2.66, bias=False, ln_type="WithBias", dual_pixel_task=False)


def weights_digest(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


def hint_case(name, cfg, seed, h, w, b=1, mod_gain=1.0):
    sd = make_hint_state(cfg, seed=seed, gain=0.6, mod_gain=mod_gain)
    x1 = synthetic_images(b, h, w, seed=seed + 10, sigma=50)
    x2 = synthetic_images(b, h, w, seed=seed + 11, sigma=50)
    taps = {}
    y1, c1 = ho.hint_forward(sd, torch.from_numpy(x1), cfg, None, tap=lambda n, t: taps.__setitem__(n, t))
    names = [n[:-len(".attn.perm")] for n in taps if n.endswith(".attn.perm")]
    perms1 = np.concatenate([taps[n + ".attn.perm"].numpy().astype(np.int32) for n in names])
    sims1 = np.concatenate([taps[n + ".attn.mean_similarity"].numpy() for n in names])
    taps2 = {}
    y2, c2 = ho.hint_forward(sd, torch.from_numpy(x2), cfg, c1, tap=lambda n, t: taps2.__setitem__(n, t))
    perms2 = np.concatenate([taps2[n + ".attn.perm"].numpy().astype(np.int32) for n in names])
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        x1=x1, y1=y1.numpy(), cache1=c1.numpy(), perms1=perms1, sims1=sims1,
                        x2=x2, y2=y2.numpy(), cache2=c2.numpy(), perms2=perms2,
                        perm_sizes=np.array([taps[n + ".attn.perm"].numel() for n in names], np.int32),
                        seed=np.int64(seed), mod_gain=np.float64(mod_gain), weights_sha256=np.array(weights_digest(sd)))
    print(name, "y1 std %.4f" % float(y1.std()), "cache2 range", float(c2.min()), float(c2.max()))


def simple_case(name, forward, sd, x, seed):
    y = forward(sd, torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, y=y, seed=np.int64(seed),
                        weights_sha256=np.array(weights_digest(sd)))
    print(name, "y std %.4f" % float(y.std()), "max|y| %.3f" % float(np.abs(y).max()))


def astv2_case(name, cfg, seed, h, w, b=1):
    sd = make_astv2_state(cfg, seed=seed, gain=0.6)
    simple_case(name, lambda s_, x_: astv2_oracle.astv2_forward(s_, x_, cfg), sd,
                synthetic_images(b, h, w, seed=seed + 10, sigma=25), seed)


def ast_case(name, cfg, seed, side, b=1):
    sd = make_ast_state(cfg, seed=seed, gain=0.5)
    simple_case(name, lambda s_, x_: ast_oracle.ast_forward(s_, x_, cfg), sd,
                synthetic_images(b, side, side, seed=seed + 10), seed)


def fpro_case(name, cfg, seed, h, w, b=1):
    sd = make_fpro_state(cfg, seed=seed, gain=0.6)
    simple_case(name, lambda s_, x_: fpro_oracle.fpro_forward(s_, x_, cfg), sd,
                synthetic_images(b, h, w, seed=seed + 10, sigma=25), seed)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    hint_case("hint_small_48x64", HINT_SMALL, 3, 48, 64)
    hint_case("hint_full_32x32", ho.default_config(), 4, 32, 32)
    # strong inter-modulators push the attention outside [0, 1): the cache leaves zero
    hint_case("hint_small_modgain_32x48", HINT_SMALL, 5, 32, 48, mod_gain=3.0)
    astv2_case("astv2_small_40x56", ASTV2_SMALL, 7, 40, 56, b=2)
    ast_case("ast_shallow_128", AST_SHALLOW, 8, 128)
    fpro_case("fpro_small_64x96", FPRO_SMALL, 9, 64, 96)
