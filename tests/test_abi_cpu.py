"""CPU tests of the boundary: the C-ABI library loads and exports what include/jirf_hip.h declares,
and the host logic (registry, key layout, aliases, error behaviour) mirrors the reference's."""
import os
import re

import numpy as np
import pytest

import jirf_hip
from oracle import hint_oracle as ho
from oracle.weights import make_hint_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "jirf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jirf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = jirf_hip.lib()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/jirf_hip.h but not exported"
        assert n in jirf_hip.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(jirf_hip.SIGNATURES) == names
    assert lib.jirf_abi_version() == 1


def test_gemm_arithmetic_switch_returns_the_previous_setting():
    """No GPU needed: the switch is host state.  The split-bf16 forms are on unless the environment says otherwise."""
    lib = jirf_hip.lib()
    default = 0 if os.environ.get("JIRF_GEMM_SPLIT_BF16", "1") in ("0", "") else 1
    before = lib.jirf_set_gemm_split_bf16(0)
    try:
        assert before == default
        assert lib.jirf_set_gemm_split_bf16(1) == 0
        assert lib.jirf_set_gemm_split_bf16(5) == 1      # any non-zero value means on
        assert lib.jirf_set_gemm_split_bf16(1) == 1
    finally:
        lib.jirf_set_gemm_split_bf16(before)


def test_hint_key_layout_matches_reference_layout():
    cfg = ho.default_config()
    net = jirf_hip.Net("HINT", cfg)
    specs = dict(net.param_specs())
    sd = make_hint_state(cfg, seed=1)
    assert set(specs) == set(sd), (sorted(set(specs) ^ set(sd))[:6])
    for k, v in sd.items():
        assert tuple(specs[k]) == tuple(v.shape), k
    assert sum(int(np.prod(s)) for s in specs.values()) == 24873812   # ~24.9 M parameters
    # a few keys spelled out against the reference constructors (hint_arch.py:304-332,156-178)
    assert specs["latent.0.attn.temperature"] == (4, 1, 1)
    assert specs["latent.0.attn.inter_modulator4.conv_width.weight"] == (288, 144, 1)
    assert specs["decoder_level1.0.ffn.project_in.weight"] == (510, 96, 1, 1)
    assert specs["encoder_level1.0.ffn.dwconv.weight"] == (254, 1, 3, 3)
    assert "encoder_level1.0.norm1.body.weight" not in specs      # is_att=False blocks have no norm1


def test_config_variants_change_the_layout():
    cfg = dict(ho.default_config(), bias=True, ln_type="BiasFree", dual_pixel_task=True)
    specs = dict(jirf_hip.Net("HINT", cfg).param_specs())
    assert "latent.0.attn.qkv.bias" in specs and "output.bias" in specs
    assert "latent.0.norm1.body.bias" not in specs
    assert specs["skip_conv.weight"] == (96, 48, 1, 1)
    assert "patch_embed.proj.bias" not in specs                    # OverlapPatchEmbed keeps bias=False


def test_registry_errors_mirror_the_reference():
    from basicsr.models.archs import define_network
    with pytest.raises(ValueError, match="Restormer is not found."):
        define_network({"type": "Restormer", "dim": 48})
    with pytest.raises(TypeError):
        define_network({"type": "HINT", "no_such_option": 1})
    with pytest.raises(ValueError, match="is not found"):
        jirf_hip.Net("NoSuchArch", {})
    with pytest.raises(TypeError):
        jirf_hip.Net("HINT", {"window": 8})
    opt = dict(ho.default_config(), type="HINT")
    net = define_network(opt)
    assert "type" not in opt and type(net).__name__ == "HINT"


def test_set_param_validation_and_aliases():
    net = jirf_hip.Net("HINT", ho.default_config())
    with pytest.raises(jirf_hip.JirfError) as e:
        net.set_param("latent.0.attn.qkv.weight", np.zeros((3, 3), np.float32))
    assert e.value.code == -1 and "shape" in str(e.value)
    with pytest.raises(jirf_hip.JirfError) as e:
        net.set_param("latent.0.attn.nope", np.zeros(3, np.float32))
    assert e.value.code == -2
    w = np.arange(288 * 144, dtype=np.float32).reshape(288, 144, 1)
    net.set_param("module.latent.0.attn.inter_modulators.3.conv_width.weight", w)      # Sequential alias + DataParallel prefix
    assert np.array_equal(net.get_param("latent.0.attn.inter_modulator4.conv_width.weight", w.shape), w)
    net.set_param("qv_cache", np.zeros((1, 96, 96), np.float32))                      # accepted and dropped
    net.set_param("latent.0.attn.temperature", np.ones(4, np.float64))                # [4] for [4,1,1], f64 converted
    with pytest.raises(jirf_hip.JirfError) as e:                                      # forward path needs finalize
        net.plan(1, 64, 64)
    assert e.value.code == -3


def test_module_state_dict_roundtrip_and_strict_load():
    from basicsr.models.archs.hint_arch import HINT
    cfg = dict(ho.default_config())
    cfg.update(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    sd = make_hint_state(cfg, seed=2)
    net = HINT(**cfg)
    net.load_state_dict(dict(sd, qv_cache=np.zeros((1, 32, 32), np.float32)))
    out = net.state_dict()
    # the reference's state dict carries `qv_cache` (and its loaders delete it): zeros when None
    assert set(out) == set(sd) | {"qv_cache"} and all(np.array_equal(out[k], sd[k]) for k in sd)
    assert out["qv_cache"].shape == (1, 32, 32) and not out["qv_cache"].any()
    bad = dict(sd)
    bad.pop("output.weight")
    with pytest.raises(KeyError):
        HINT(**cfg).load_state_dict(bad)
    with pytest.raises(KeyError):
        HINT(**cfg).load_state_dict(dict(sd, extra=np.zeros(1)))
    with pytest.raises(ValueError):
        HINT(**cfg).load_state_dict(dict(sd, **{"output.weight": np.zeros((3, 3))}))
    assert net.qv_cache is None and net.eval() is net and net.cuda() is net
    with pytest.raises(ValueError):
        net._check_input((1, 3, 30, 32))


def test_no_cpu_fallback():
    """Without a GPU the compute path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = jirf_hip.Net("HINT", ho.default_config())
    for k, v in make_hint_state(ho.default_config(), seed=1).items():
        net.set_param(k, v)
    with pytest.raises(jirf_hip.JirfError) as e:
        net.finalize()
    assert e.value.code == -6


def test_astv2_and_ast_key_layouts():
    from oracle import ast_oracle, astv2_oracle
    from oracle.weights import make_ast_state, make_astv2_state
    specs = dict(jirf_hip.Net("ASTv2", astv2_oracle.default_config()).param_specs())
    sd = make_astv2_state()
    assert set(specs) == set(sd) and all(tuple(specs[k]) == sd[k].shape for k in sd)
    assert specs["decoder_level3.0.attn.att_ca.weight"] == (48, 24, 1, 1)          # 2ch x ch, ch = 192 / 8
    assert specs["encoder_level1.0.ffn.partial_conv3.weight"] == (2, 1, 3, 3)       # dim // 24
    assert specs["decoder_level1.0.attn.temperature"] == (2, 1, 1)

    specs = dict(jirf_hip.Net("AST", ast_oracle.default_config()).param_specs())
    sd = make_ast_state()
    buffers = {k for k in sd if k.endswith("relative_position_index")}
    assert set(specs) == set(sd) - buffers and all(tuple(specs[k]) == sd[k].shape for k in specs)
    assert sum(int(np.prod(s)) for s in specs.values()) == 26339347 - 10 * 64 * 64  # ~26.3 M parameters
    assert specs["conv.blocks.0.attn.relative_position_bias_table"] == (225, 16)
    assert specs["upsample_1.deconv.0.weight"] == (512, 128, 2, 2)
    assert specs["dowsample_0.conv.0.weight"] == (64, 32, 4, 4)
    assert "encoderlayer_0.blocks.0.norm1.weight" not in specs                      # att=False blocks
    net = jirf_hip.Net("AST", ast_oracle.default_config())
    net.set_param("conv.blocks.0.attn.relative_position_index", np.zeros((64, 64), np.int32))   # accepted, recomputed


def test_ast_constructor_contract():
    from basicsr.models.archs import define_network
    net = define_network(dict(type="AST", token_mlp="frfn", some_unknown_option=1))   # **kwargs is swallowed
    assert type(net).__name__ == "AST"
    leff = define_network(dict(type="AST"))                                           # default token_mlp="leff"
    specs = dict(leff._specs)
    assert not any("partial_conv3" in k for k in specs)
    assert tuple(specs["encoderlayer_0.blocks.0.mlp.linear1.0.weight"]) == (128, 32)
    plain = define_network(dict(type="AST", token_mlp="ffn"))
    assert "conv.blocks.1.mlp.fc2.bias" in dict(plain._specs)
    with pytest.raises(Exception, match="FFN error"):
        define_network(dict(type="AST", token_mlp="nope"))
    with pytest.raises(TypeError):
        define_network(dict(type="ASTv2", some_unknown_option=1))                     # no **kwargs there
