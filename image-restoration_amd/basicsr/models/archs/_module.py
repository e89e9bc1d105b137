"""Host-side mirror of the `nn.Module` surface the reference's scripts use on a network_g:
`load_state_dict`, `state_dict`, `cuda`, `eval`, `train`, `__call__` (e.g.
Deraining/test_rain100L.py:117-155 of the reference).  All arithmetic happens in
libjirf_hip.so; this class keeps the host copy of the parameters and marshals tensors."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import numpy as np

import jirf_hip

try:  # torch is plumbing only (device memory + streams)
    import torch
except Exception:  # pragma: no cover
    torch = None


def _to_numpy(v) -> np.ndarray:
    if isinstance(v, np.ndarray):
        return v
    if torch is not None and isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    if hasattr(v, "numpy"):
        return np.asarray(v.numpy())
    return np.asarray(v)


class JirfModule:
    ARCH = ""
    # semantics of a B > 1 call: "coupled" is the reference's literal batched call, "per_image"
    # equals B separate B=1 calls (what every test script does)
    batch_mode = "coupled"

    def __init__(self, cfg: dict, device: int = 0):
        self._cfg = dict(cfg)
        self._device = device
        self._spec_net = jirf_hip.Net(self.ARCH, self._cfg, device)   # host-only until finalize
        self._specs = OrderedDict(self._spec_net.param_specs())
        self._params: "OrderedDict[str, np.ndarray]" = OrderedDict()
        self._net: Optional[jirf_hip.Net] = None
        self.training = False

    # ---- nn.Module surface ------------------------------------------------------------------------
    def named_parameter_shapes(self) -> "OrderedDict[str, tuple]":
        return OrderedDict(self._specs)

    def state_dict(self) -> "OrderedDict[str, np.ndarray]":
        return OrderedDict((k, self._params[k].copy()) for k in self._specs if k in self._params)

    def _canonical(self, key: str) -> Optional[str]:
        return key[7:] if key.startswith("module.") else key

    def load_state_dict(self, state: Dict[str, object], strict: bool = True):
        new = OrderedDict()
        unexpected = []
        for k, v in state.items():
            ck = self._canonical(k)
            if ck is None:
                continue
            if ck not in self._specs:
                unexpected.append(k)
                continue
            a = np.ascontiguousarray(_to_numpy(v), dtype=np.float32)
            want = self._specs[ck]
            if tuple(d for d in a.shape if d != 1) != tuple(d for d in want if d != 1):
                raise ValueError(f"size mismatch for {k}: got {tuple(a.shape)}, expected {tuple(want)}")
            new[ck] = a.reshape(want)
        missing = [k for k in self._specs if k not in new and k not in self._params]
        if strict and (missing or unexpected):
            raise KeyError(f"load_state_dict: missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                           f"unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        self._params.update(new)
        self._drop_native()
        return self

    def cuda(self, device: Optional[int] = None):
        """`.cuda()` with no argument binds to torch's current device, like nn.Module.cuda."""
        if device is None and torch is not None and torch.cuda.is_available():
            device = torch.cuda.current_device()
        if device is not None and not isinstance(device, int):
            device = getattr(device, "index", None) if not isinstance(device, str) else torch.device(device).index
            device = 0 if device is None else device
        if device is not None and int(device) != self._device:
            self._device = int(device)
            self._drop_native()
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("libjirf_hip provides the inference forward only")
        return self.eval()

    def parameters(self):
        return [self._params[k] for k in self._specs if k in self._params]

    # ---- native net lifecycle ---------------------------------------------------------------------
    def _drop_native(self):
        if self._net is not None:
            self._on_drop_native()
            self._net.close()
            self._net = None

    def _on_drop_native(self):
        pass

    def _on_new_native(self):
        pass

    def native(self) -> jirf_hip.Net:
        if self._net is None:
            missing = [k for k in self._specs if k not in self._params]
            if missing:
                raise RuntimeError(f"{self.ARCH}: parameters were never loaded: {missing[:5]}"
                                   f"{'...' if len(missing) > 5 else ''}")
            net = jirf_hip.Net(self.ARCH, self._cfg, self._device)
            for k in self._specs:
                net.set_param(k, self._params[k])
            net.finalize()
            self._net = net
            self._on_new_native()
        return self._net

    # ---- forward ------------------------------------------------------------------------------------
    def _mode(self) -> int:
        return jirf_hip.BATCH_COUPLED if self.batch_mode == "coupled" else jirf_hip.BATCH_PER_IMAGE

    def _check_input(self, shape):
        if len(shape) != 4:
            raise ValueError(f"{self.ARCH} expects an NCHW tensor, got shape {tuple(shape)}")

    def _out_channels(self, in_channels: int) -> int:
        return in_channels

    def __call__(self, x):
        return self.forward(x)

    def execute(self, x):  # the reference's spelling of forward
        return self.forward(x)

    def forward(self, x):
        net = self.native()
        self._check_input(x.shape)
        b, c, h, w = (int(d) for d in x.shape)
        oc = self._out_channels(c)
        if torch is not None and isinstance(x, torch.Tensor):
            if x.is_cuda:
                if x.device.index != self._device:
                    raise ValueError(f"{self.ARCH}: input lives on cuda:{x.device.index} but the network is bound "
                                     f"to cuda:{self._device} (move it with .cuda({x.device.index}))")
                xc = x.detach().contiguous().float()
                y = torch.empty((b, oc, h, w), dtype=torch.float32, device=x.device)
                stream = torch.cuda.current_stream(x.device).cuda_stream
                net.forward_ptr(xc.data_ptr(), y.data_ptr(), b, h, w, self._mode(), stream)
                return y
            y = net.forward_host(x.detach().numpy(), self._mode(), oc)
            return torch.from_numpy(y)
        return net.forward_host(_to_numpy(x), self._mode(), oc)
