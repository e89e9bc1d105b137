"""ASTv2 network_g class backed by libjirf_hip.so.

Same constructor keywords, state-dict keys/shapes and call contract as the reference class
(basicsr/models/archs/astv2_arch.py:302-469 of the reference); H and W must be multiples of 4
(two stride-2 levels; the test scripts pad to 8).
"""
from __future__ import annotations

from ._module import JirfModule


class ASTv2(JirfModule):
    ARCH = "ASTv2"

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
        if shape[2] % 4 or shape[3] % 4:
            raise ValueError(f"ASTv2 needs H and W to be multiples of 4, got {shape[2]}x{shape[3]}")
