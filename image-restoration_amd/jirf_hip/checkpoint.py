"""Checkpoint files of the reference (`net_g_<iter>.jth`) and the loader rules of its model class.

The reference writes `{"params": state_dict [, "params_ema": state_dict]}` with `jt.save`
(basicsr/models/base_model.py:176-193) and reads it back with `jt.load` followed by the key rules of
`load_network` (base_model.py:228-258).  `jt.save` serialises with `pickle` after turning every array
into a numpy array, then appends the 20-byte SHA-1 of the pickle and the 8-byte mark ``HCAJSLHD``.

This module reads and writes that container without Jittor.  Reading is done by an unpickler that only
rebuilds dicts, lists, scalars and numpy arrays and refuses every other global, so a checkpoint cannot run
code when it is opened.  `.pth` / `.pt` zip archives go through `torch.load(weights_only=True)`.
"""
from __future__ import annotations

import hashlib
import io
import logging
import pickle
import zipfile
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np

logger = logging.getLogger("basicsr")

TRAILER_MARK = b"HCAJSLHD"
_TRAILER_LEN = 20 + len(TRAILER_MARK)

# the globals a pickled tree of numpy arrays refers to (numpy 1.x spells the module numpy.core, 2.x numpy._core)
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"),
    ("numpy", "ndarray"),
    ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"),
}


class CheckpointError(ValueError):
    pass


class _ArrayOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in _ALLOWED_GLOBALS:
            raise CheckpointError(f"checkpoint refers to {module}.{name}; only dicts, lists, scalars and numpy arrays "
                                  "are accepted")
        if module.startswith("numpy.core"):
            module = module.replace("numpy.core", "numpy._core" if hasattr(np, "_core") else "numpy.core", 1)
        return super().find_class(module, name)


def _to_numpy_tree(obj):
    if isinstance(obj, dict):
        return type(obj)((k, _to_numpy_tree(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_numpy_tree(v) for v in obj)
    if isinstance(obj, (np.ndarray, np.generic, int, float, str, bool, bytes)) or obj is None:
        return obj
    if hasattr(obj, "detach"):            # torch.Tensor
        return obj.detach().cpu().numpy()
    if hasattr(obj, "numpy"):
        return np.asarray(obj.numpy())
    raise CheckpointError(f"cannot store an object of type {type(obj).__name__} in a checkpoint")


def dumps(obj) -> bytes:
    body = pickle.dumps(_to_numpy_tree(obj), protocol=4)
    return body + hashlib.sha1(body).digest() + TRAILER_MARK


def loads(data: bytes):
    """Inverse of `dumps`.  A file without the trailer (a bare pickle) is accepted; a trailer whose digest
    does not match the body is an error."""
    if len(data) >= _TRAILER_LEN and data[-len(TRAILER_MARK):] == TRAILER_MARK:
        body, digest = data[:-_TRAILER_LEN], data[-_TRAILER_LEN:-len(TRAILER_MARK)]
        if hashlib.sha1(body).digest() != digest:
            raise CheckpointError("checkpoint is damaged: SHA-1 of the body does not match its trailer")
    else:
        body = data
    try:
        return _ArrayOnlyUnpickler(io.BytesIO(body)).load()
    except pickle.UnpicklingError as e:
        raise CheckpointError(f"not a checkpoint this loader understands: {e}") from e


def save(obj, path: str) -> None:
    """`jt.save(obj, path)` (base_model.py:193)."""
    with open(path, "wb") as f:
        f.write(dumps(obj))


def load(path: str):
    """`jt.load(path)` (base_model.py:241; every `*/test_*.py`, e.g. Deraining/test_rain100L.py:119)."""
    if zipfile.is_zipfile(path):           # torch's archive format (`*.pth` weights named by some scripts)
        import torch
        tree = torch.load(path, map_location="cpu", weights_only=True)
        return _to_numpy_tree(tree)
    with open(path, "rb") as f:
        return loads(f.read())


def select_params(load_net: Dict, arch_name: str, param_key: Optional[str] = "params") -> "OrderedDict[str, np.ndarray]":
    """The key rules of `load_network` (base_model.py:243-256) on an opened checkpoint: HINT files lose
    `params.qv_cache`; a missing `param_key` falls back to `params`; a leading `module.` is stripped."""
    if arch_name == "HINT" and isinstance(load_net.get("params"), dict):
        load_net["params"].pop("qv_cache", None)
    if param_key is not None:
        if param_key not in load_net and "params" in load_net:
            logger.info("Loading: %s does not exist, use params.", param_key)
            param_key = "params"
        if param_key not in load_net:
            raise KeyError(f"checkpoint has no '{param_key}' entry (top-level keys: {list(load_net)[:8]})")
        load_net = load_net[param_key]
    out = OrderedDict()
    for k, v in load_net.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def load_network(net, load_path: str, strict: bool = True, param_key: Optional[str] = "params"):
    """`BaseModel.load_network` (base_model.py:228-258) for a network_g of this package.  As there, key
    mismatches are reported (`_print_different_keys_loading`, base_model.py:195-226) and the keys both
    sides know are loaded: the reference calls `net.load_state_dict(load_net)` without a strict flag, and
    `strict` only decides whether entries of a different size are set aside.  A parameter that no file
    ever supplied still fails, by name, when the network is first used."""
    logger.info("Loading %s model from %s.", net.__class__.__name__, load_path)
    state = select_params(load(load_path), net.__class__.__name__, param_key)
    shapes = net.named_parameter_shapes()
    canon = getattr(net, "_canonical", lambda k: k)
    have = {canon(k) for k in state} - {None}
    missing = sorted(set(shapes) - have)
    extra = sorted(k for k in state if canon(k) is not None and canon(k) not in shapes)
    if missing or extra:
        logger.warning("Current net - loaded net:")
        for k in missing:
            logger.warning("  %s", k)
        logger.warning("Loaded net - current net:")
        for k in extra:
            logger.warning("  %s", k)
    if not strict:     # base_model.py:221-226: entries whose size differs are set aside, not loaded
        for k in list(state):
            ck = canon(k)
            if ck in shapes and int(np.prod(np.shape(state[k]))) != int(np.prod(shapes[ck])):
                logger.warning("Size different, ignore [%s]: crt_net: %s; load_net: %s", k, shapes[ck], np.shape(state[k]))
                state[k + ".ignore"] = state.pop(k)
    net.load_state_dict(state, strict=False)
    return net


def save_network(net, save_path: str, param_key="params") -> None:
    """`BaseModel.save_network` (base_model.py:176-193): one entry per (net, param_key) pair."""
    nets = net if isinstance(net, list) else [net]
    keys = param_key if isinstance(param_key, list) else [param_key]
    if len(nets) != len(keys):
        raise ValueError("save_network: one param_key per network")
    save({k: n.state_dict() for n, k in zip(nets, keys)}, save_path)
