"""FPro network_g class backed by libjirf_hip.so.

Same constructor keywords, state-dict keys/shapes and call contract as the reference class
(basicsr/models/archs/fpro_arch.py:548-718 of the reference).  H and W must be multiples of 32
(two stride-2 levels and 8x8 windows at the coarsest prompt; the test scripts feed 256x256 tiles).
"""
from __future__ import annotations

from ._module import JirfModule


class FPro(JirfModule):
    ARCH = "FPro"

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8],
                 num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                 ln_type="WithBias", dual_pixel_task=False):
        cfg = dict(inp_channels=inp_channels, out_channels=out_channels, dim=dim,
                   num_blocks=list(num_blocks), num_refinement_blocks=num_refinement_blocks,
                   heads=list(heads), ffn_expansion_factor=ffn_expansion_factor, bias=bool(bias),
                   ln_type=ln_type, dual_pixel_task=bool(dual_pixel_task))
        super().__init__(cfg)
        self._out_ch = out_channels

    def _out_channels(self, in_channels: int) -> int:
        return self._out_ch

    def _check_input(self, shape):
        super()._check_input(shape)
        if shape[2] % 32 or shape[3] % 32:
            raise ValueError(f"FPro needs H and W to be multiples of 32, got {shape[2]}x{shape[3]}")
