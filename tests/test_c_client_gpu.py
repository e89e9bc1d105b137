"""The C ABI used from plain C: tools/jirf_kat (built by __graft_entry__.build() from tools/jirf_kat.c, no
Python and no HIP calls in it) runs the committed golden cases through libjirf_hip.so with host buffers."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from jirf_hip import synthetic
from tests.make_golden import AST_SHALLOW, ASTV2_SMALL, HINT_SMALL, weights_digest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "tools", "jirf_kat")


def _write_case(path, arch, cfg, sd, x, y, perms=None, mode=1, tol=1e-3):
    def s(b):
        return struct.pack("<I", len(b)) + b
    with open(path, "wb") as f:
        f.write(b"JKAT1\0\0\0")
        f.write(s(arch.encode()))
        f.write(s(json.dumps(cfg).encode()))
        B, Cin, H, W = x.shape
        f.write(struct.pack("<6If", mode, B, H, W, Cin, y.shape[1], tol))
        f.write(struct.pack("<I", len(sd)))
        for k, v in sd.items():
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            f.write(s(k.encode()))
            f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())
        p = np.zeros(0, np.int32) if perms is None else np.ascontiguousarray(perms, dtype=np.int32).ravel()
        f.write(struct.pack("<Q", p.size))
        f.write(p.tobytes())
        f.write(np.ascontiguousarray(x, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(y, dtype=np.float32).tobytes())


def _run(case):
    if not os.path.exists(BIN):
        pytest.fail("tools/jirf_kat is not built: run __graft_entry__.build()")
    r = subprocess.run([BIN, case], capture_output=True, text=True, timeout=100)
    print(r.stdout, r.stderr)
    return r


def test_hint_golden_through_plain_c(tmp_path):
    g = np.load(os.path.join(GOLD, "hint_small_48x64.npz"))
    sd = synthetic.make_hint_state(HINT_SMALL, seed=int(g["seed"]), gain=0.6, mod_gain=float(g["mod_gain"]))
    assert weights_digest(sd) == str(g["weights_sha256"])
    case = str(tmp_path / "hint.jkat")
    _write_case(case, "HINT", HINT_SMALL, sd, g["x1"], g["y1"], perms=g["perms1"])
    r = _run(case)
    assert r.returncode == 0 and "max_abs_err" in r.stdout
    # a wrong expectation must be reported, not passed
    _write_case(case, "HINT", HINT_SMALL, sd, g["x1"], g["y1"] + 0.01, perms=g["perms1"])
    assert _run(case).returncode == 1


def test_astv2_and_ast_golden_through_plain_c(tmp_path):
    g = np.load(os.path.join(GOLD, "astv2_small_40x56.npz"))
    sd = synthetic.make_astv2_state(ASTV2_SMALL, seed=int(g["seed"]), gain=0.6)
    case = str(tmp_path / "astv2.jkat")
    _write_case(case, "ASTv2", ASTV2_SMALL, sd, g["x"], g["y"])
    assert _run(case).returncode == 0
    g = np.load(os.path.join(GOLD, "ast_shallow_128.npz"))
    sd = synthetic.make_ast_state(AST_SHALLOW, seed=int(g["seed"]), gain=0.5)
    case = str(tmp_path / "ast.jkat")
    _write_case(case, "AST", AST_SHALLOW, sd, g["x"], g["y"])
    assert _run(case).returncode == 0


def test_errors_surface_in_c(tmp_path):
    g = np.load(os.path.join(GOLD, "astv2_small_40x56.npz"))
    sd = synthetic.make_astv2_state(ASTV2_SMALL, seed=int(g["seed"]), gain=0.6)
    sd.pop(next(iter(sd)))                                   # a missing parameter: finalize must refuse
    case = str(tmp_path / "bad.jkat")
    _write_case(case, "ASTv2", ASTV2_SMALL, sd, g["x"], g["y"])
    r = _run(case)
    assert r.returncode == 2 and "jirf_net_finalize" in r.stderr
