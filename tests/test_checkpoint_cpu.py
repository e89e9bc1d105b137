"""Checkpoint container and the loader rules of the reference's model class
(basicsr/models/base_model.py:176-258), without a GPU."""
import hashlib
import pickle

import numpy as np
import pytest

from jirf_hip import checkpoint as ck
from jirf_hip import synthetic


class _FakeNet:
    def __init__(self, shapes, name="HINT"):
        self._shapes = shapes
        self.loaded = None
        self.__class__ = type(name, (_FakeNet,), {})

    def named_parameter_shapes(self):
        return dict(self._shapes)

    def load_state_dict(self, state, strict=True):
        self.loaded = (dict(state), strict)

    def state_dict(self):
        return {k: np.zeros(s, np.float32) for k, s in self._shapes.items()}


def test_round_trip_and_trailer(tmp_path):
    sd = {"a.weight": np.arange(12, dtype=np.float32).reshape(3, 4), "b.index": np.arange(5, dtype=np.int32)}
    p = tmp_path / "net_g_latest.jth"
    ck.save({"params": sd}, str(p))
    raw = p.read_bytes()
    assert raw.endswith(ck.TRAILER_MARK)
    assert raw[-28:-8] == hashlib.sha1(raw[:-28]).digest()
    assert pickle.loads(raw[:-28])["params"]["a.weight"].dtype == np.float32      # plain pickle reads the body
    back = ck.load(str(p))
    for k in sd:
        np.testing.assert_array_equal(back["params"][k], sd[k])
        assert back["params"][k].dtype == sd[k].dtype


def test_bare_pickle_and_damage(tmp_path):
    sd = {"params": {"w": np.ones((2, 2), np.float32)}}
    p = tmp_path / "bare.pkl"
    p.write_bytes(pickle.dumps(sd, protocol=4))
    np.testing.assert_array_equal(ck.load(str(p))["params"]["w"], sd["params"]["w"])
    raw = bytearray(ck.dumps(sd))
    raw[10] ^= 0xFF
    with pytest.raises(ck.CheckpointError, match="SHA-1"):
        ck.loads(bytes(raw))


def test_refuses_other_globals():
    evil = pickle.dumps({"params": {"w": print}}, protocol=4)
    with pytest.raises(ck.CheckpointError, match="builtins.print"):
        ck.loads(evil)


def test_torch_tensors_are_stored_as_numpy(tmp_path):
    import torch
    p = tmp_path / "t.jth"
    ck.save({"params": {"w": torch.arange(6.0).reshape(2, 3)}}, str(p))
    w = ck.load(str(p))["params"]["w"]
    assert isinstance(w, np.ndarray) and w.shape == (2, 3)
    t = tmp_path / "t.pth"
    torch.save({"params": {"w": torch.arange(6.0).reshape(2, 3)}}, str(t))
    np.testing.assert_array_equal(ck.load(str(t))["params"]["w"], w)


def test_select_params_rules():
    tree = {"params": {"module.x": np.zeros(1), "y": np.ones(1), "qv_cache": np.zeros((1, 96, 96))}}
    out = ck.select_params(tree, "HINT", "params_ema")              # falls back to params, drops qv_cache
    assert list(out) == ["x", "y"]
    tree = {"params": {"qv_cache": np.zeros(1), "y": np.ones(1)}, "params_ema": {"y": np.full(1, 2.0)}}
    assert ck.select_params(tree, "ASTv2", "params_ema")["y"][0] == 2.0
    assert "qv_cache" in ck.select_params({"params": {"qv_cache": np.zeros(1)}}, "ASTv2")
    assert list(ck.select_params({"w": np.zeros(1)}, "AST", None)) == ["w"]
    with pytest.raises(KeyError):
        ck.select_params({"other": {}}, "AST", "params")


def test_load_network_non_strict_sets_mismatches_aside(tmp_path):
    p = tmp_path / "n.jth"
    ck.save({"params": {"a": np.zeros((3, 4), np.float32), "b": np.zeros(7, np.float32)}}, str(p))
    net = _FakeNet({"a": (3, 4), "b": (5,)}, "ASTv2")
    ck.load_network(net, str(p), strict=False)
    state, strict = net.loaded
    assert strict is False and set(state) == {"a", "b.ignore"}


def test_full_state_dict_round_trip(tmp_path):
    cfg = synthetic.default_config("ASTv2")
    sd = synthetic.make_state("ASTv2", cfg, seed=3)
    p = tmp_path / "net_g_1000.jth"
    ck.save({"params": sd, "params_ema": sd}, str(p))
    back = ck.select_params(ck.load(str(p)), "ASTv2", "params_ema")
    assert list(back) == list(sd)
    for k, v in sd.items():
        np.testing.assert_array_equal(back[k], np.asarray(v))


def test_load_network_into_the_hint_class(tmp_path):
    """The file a training run leaves (`params` with `qv_cache` and DataParallel's `module.` prefix) loads
    into the HINT class through the rules of base_model.py:243-256 and is written back key for key."""
    from basicsr.models.archs.hint_arch import HINT
    cfg = dict(synthetic.default_config("HINT"))
    cfg.update(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    sd = synthetic.make_state("HINT", cfg, seed=2)
    on_disk = {"module." + k: v for k, v in sd.items()}
    on_disk["qv_cache"] = np.zeros((1, 32, 32), np.float32)
    p = tmp_path / "net_g_latest.jth"
    ck.save({"params": on_disk}, str(p))
    net = ck.load_network(HINT(**cfg), str(p), strict=True, param_key="params_ema")
    out = net.state_dict()
    assert set(out) == set(sd) | {"qv_cache"} and all(np.array_equal(out[k], np.asarray(sd[k])) for k in sd)
    q = tmp_path / "copy.jth"
    ck.save_network(net, str(q))
    again = ck.load(str(q))["params"]
    assert list(again) == list(out)
    # the reference's loader and every HINT test script run `del checkpoint["params"]["qv_cache"]` without a
    # guard (base_model.py:243-244, Deraining/test_rain100L.py:121-122): a file this package writes must hold it
    assert again["qv_cache"].shape == (1, 32, 32) and again["qv_cache"].dtype == np.float32
    del again["qv_cache"]
    assert set(again) == set(sd)


def test_load_network_reports_key_mismatches_and_loads_the_rest(tmp_path, caplog):
    """base_model.py:257-258: differing keys are logged, then `net.load_state_dict(load_net)` runs without a
    strict flag - an extra or missing entry does not abort the load."""
    from basicsr.models.archs.hint_arch import HINT
    cfg = dict(synthetic.default_config("HINT"))
    cfg.update(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    sd = dict(synthetic.make_state("HINT", cfg, seed=2))
    dropped = "output.weight"
    kept = {k: v for k, v in sd.items() if k != dropped}
    kept["some_buffer_the_net_does_not_have"] = np.zeros(3, np.float32)
    kept["qv_cache"] = np.zeros((1, 32, 32), np.float32)
    p = tmp_path / "partial.jth"
    ck.save({"params": kept}, str(p))
    import logging
    with caplog.at_level(logging.WARNING, logger="basicsr"):
        net = ck.load_network(HINT(**cfg), str(p), strict=True)
    text = caplog.text
    assert dropped in text and "some_buffer_the_net_does_not_have" in text
    out = net.state_dict()
    assert dropped not in out and "patch_embed.proj.weight" in out
