"""HINT network_g class backed by libjirf_hip.so.

Same constructor keywords, state-dict keys/shapes and call contract as the
reference class (basicsr/models/archs/hint_arch.py:494-706 of the reference):
`HINT(**yaml_network_g)`, `load_state_dict`, `cuda()`, `eval()`, `net(x)` with
x an NCHW fp32 tensor whose H and W are multiples of 8.  Like the reference
module, the instance carries `qv_cache` from call to call (reference :511,
:660-696); a freshly constructed instance starts with `qv_cache = None`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ._module import JirfModule, _to_numpy, torch


class HINT(JirfModule):
    ARCH = "HINT"

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                 num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                 ln_type="WithBias", dual_pixel_task=False, qv_cache=None):
        cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim,
                   num_blocks=list(num_blocks), num_refinement_blocks=num_refinement_blocks,
                   heads=list(heads), ffn_expansion_factor=ffn_expansion_factor, bias=bool(bias),
                   ln_type=ln_type, dual_pixel_task=bool(dual_pixel_task))
        super().__init__(cfg)
        self._out_ch = out_channels
        self._pending_cache = None if qv_cache is None else np.asarray(_to_numpy(qv_cache), np.float32)
        self._cache_dirty = qv_cache is not None

    def _canonical(self, key: str) -> Optional[str]:
        key = super()._canonical(key)
        if key == "qv_cache":       # saved checkpoints hold it; every test script deletes it
            return None
        tag = ".attn.inter_modulators."
        pos = key.find(tag)
        if pos >= 0:                # the Sequential alias of inter_modulator1..4 (reference :325-332)
            d = pos + len(tag)
            key = f"{key[:pos]}.attn.inter_modulator{int(key[d]) + 1}{key[d + 1:]}"
        return key

    def _out_channels(self, in_channels: int) -> int:
        return self._out_ch

    def state_dict(self):
        """The reference module keeps `qv_cache` as an attribute that its state dict carries (saved files
        hold `params.qv_cache`, and base_model.py:243-244 plus every HINT test script delete that key
        without a guard), so a file written from this dict must have it: the carried cache (batch mean
        when it is per image), or zeros [1, C, C] of the last stage's width when it is None."""
        sd = super().state_dict()
        cache = self.qv_cache
        if cache is None:
            c = 2 * int(self._cfg["dim"])
            cache = np.zeros((1, c, c), np.float32)
        elif cache.shape[0] != 1:
            cache = cache.mean(axis=0, keepdims=True).astype(np.float32)
        sd["qv_cache"] = np.ascontiguousarray(cache, dtype=np.float32)
        return sd

    def _check_input(self, shape):
        super()._check_input(shape)
        if shape[2] % 8 or shape[3] % 8:
            raise ValueError(f"HINT needs H and W to be multiples of 8, got {shape[2]}x{shape[3]}")

    # ---- qv_cache attribute ---------------------------------------------------------------------
    @property
    def qv_cache(self):
        if self._net is None or self._cache_dirty:
            return self._pending_cache
        return self._net.get_cache()

    @qv_cache.setter
    def qv_cache(self, value):
        self._pending_cache = None if value is None else np.asarray(_to_numpy(value), np.float32)
        self._cache_dirty = True

    def _on_drop_native(self):
        if not self._cache_dirty:
            self._pending_cache = self._net.get_cache()
            self._cache_dirty = True

    def forward(self, x):
        net = self.native()
        if self._cache_dirty:
            net.set_cache(self._pending_cache)
            self._cache_dirty = False
            self._pending_cache = None
        return super().forward(x)

    # diagnostics: channel orders of the last call / pin them for the next calls
    def last_perms(self, groups: int = 1):
        return self.native().get_perms(groups)

    def force_perms(self, perms):
        self.native().set_forced_perms(perms)
