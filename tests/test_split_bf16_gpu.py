"""The split-bf16 GEMMs (the default; JIRF_GEMM_SPLIT_BF16=0 switches them off) must be as accurate as the fp32-MFMA GEMMs.

Both kernels are run in child processes (the switch is read once per process) on the same seeded
operands and compared with a float64 product; the split form keeps six of the nine bf16 partial products
of every fp32 pair and accumulates in fp32, so its error has to stay within a small factor of the fp32
kernel's own rounding error."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(split):
    path = os.pathsep.join([ROOT, os.path.join(ROOT, "image-restoration_amd"), os.environ.get("PYTHONPATH", "")])
    # JIRF_OP_LN_FOLD: the LayerNorm weight and bias folded into the GEMM weights, the form the networks run (the
    # split kernel only normalises)
    env = dict(os.environ, JIRF_GEMM_SPLIT_BF16="1" if split else "0", JIRF_OP_LN_FOLD="1", PYTHONPATH=path)
    r = subprocess.run([sys.executable, "-m", "tests.split_bf16_probe"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_split_bf16_matches_fp32_accuracy():
    fp32, split = _probe(False), _probe(True)
    assert fp32.keys() == split.keys()
    for k in fp32:
        # both are a few ulp of the output magnitude (|y| ~ 5); the absolute floor covers tiny fp32 errors
        assert split[k] <= 3.0 * fp32[k] + 2e-6, (k, fp32[k], split[k])
        assert split[k] < 5e-5, (k, split[k])


def test_whole_network_agrees_with_the_fp32_matrix_instruction():
    """HINT (the golden fixture's small configuration, channel orders pinned to the fixture's: the argsort behind them
    is discontinuous) and AST, each built once per arithmetic through jirf_set_gemm_split_bf16: the outputs differ by
    fp32 rounding only."""
    import numpy as np
    import torch

    import jirf_hip
    from basicsr.models.archs import define_network
    from jirf_hip.synthetic import default_config, make_state, synthetic_images

    L = jirf_hip.lib()
    before = L.jirf_set_gemm_split_bf16(1)
    try:
        for arch, B, S in (("HINT", 2, 128), ("AST", 1, 128)):
            cfg = default_config(arch)
            sd = make_state(arch, cfg, seed=3)
            x = torch.from_numpy(synthetic_images(B, S, S, seed=5, sigma=25)).cuda()
            outs, perms = [], None
            for split in (1, 0):
                L.jirf_set_gemm_split_bf16(split)
                net = define_network(dict(cfg, type=arch))
                net.load_state_dict(sd)
                net.cuda(0).eval()
                if arch == "HINT":
                    net.batch_mode = "per_image"
                    if perms is not None:
                        net.native().plan(B, S, S, jirf_hip.BATCH_PER_IMAGE)   # pinned orders belong to a planned shape
                        net.force_perms(perms)
                outs.append(net(x).cpu().numpy())
                if arch == "HINT" and perms is None:
                    perms = net.last_perms(groups=B)
                del net
            assert np.isfinite(outs[0]).all()
            assert np.abs(outs[0] - outs[1]).max() < 2e-5, (arch, np.abs(outs[0] - outs[1]).max())
    finally:
        L.jirf_set_gemm_split_bf16(before)
