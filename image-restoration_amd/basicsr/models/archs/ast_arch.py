"""AST network_g class backed by libjirf_hip.so.

Same constructor keywords, state-dict keys/shapes and call contract as the reference class
(basicsr/models/archs/ast_arch.py:940-1256 of the reference), every `token_mlp` the reference takes
("frfn" as shipped, "leff" the constructor default, "ffn" / "mlp"); 8x8 windows, head_dim 32.  Like the reference, the
input must be square with a side that is a multiple of 128 (every module derives
H = W = int(sqrt(L)), and 8x8 windows must tile the 1/16-resolution bottleneck).
"""
from __future__ import annotations

from typing import Optional

from ._module import JirfModule


class AST(JirfModule):
    ARCH = "AST"

    def __init__(self, img_size=256, in_chans=3, dd_in=3, embed_dim=32, depths=[2, 2, 2, 2, 2, 2, 2, 2, 2],
                 num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=None, patch_norm=True,
                 use_checkpoint=False, token_projection="linear", token_mlp="leff", dowsample=None, upsample=None,
                 shift_flag=True, **kwargs):
        # dropout / drop-path / checkpointing are identities at inference; **kwargs is swallowed like
        # the reference constructor does
        cfg = dict(img_size=img_size, in_chans=in_chans, dd_in=dd_in, embed_dim=embed_dim, depths=list(depths),
                   num_heads=list(num_heads), win_size=win_size, mlp_ratio=mlp_ratio, qkv_bias=bool(qkv_bias),
                   qk_scale=qk_scale, token_projection=token_projection, token_mlp=token_mlp,
                   shift_flag=bool(shift_flag))
        if token_mlp not in ("frfn", "leff", "ffn", "mlp"):
            raise Exception("FFN error!")
        super().__init__(cfg)
        self._out_ch = in_chans

    def _canonical(self, key: str) -> Optional[str]:
        key = super()._canonical(key)
        if key.endswith(".attn.relative_position_index"):   # a buffer the kernels recompute
            return None
        return key

    def _out_channels(self, in_channels: int) -> int:
        return self._out_ch

    def _check_input(self, shape):
        super()._check_input(shape)
        if shape[2] != shape[3] or shape[2] % 128:
            raise ValueError(f"AST needs a square input whose side is a multiple of 128, got {shape[2]}x{shape[3]}")
