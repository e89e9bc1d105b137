"""ctypes binding of libjirf_hip.so (the C ABI declared in include/jirf_hip.h).

The library is the product; this module only marshals pointers.  There is no
CPU fallback: importing works without a GPU (so symbol checks can run), but any
compute call raises if the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JIRF_HIP_LIB", os.path.join(os.path.dirname(_HERE), "libjirf_hip.so"))

BATCH_PER_IMAGE = 0
BATCH_COUPLED = 1

_F32, _I32, _I64, _F64 = 0, 1, 2, 3
_DTYPES = {np.dtype(np.float32): _F32, np.dtype(np.int32): _I32, np.dtype(np.int64): _I64,
           np.dtype(np.float64): _F64}


class JirfError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libjirf_hip error {code}: {message}")
        self.code = code


_lib = None

# name -> (restype, argtypes); must list every symbol include/jirf_hip.h declares
_vp, _i, _sz, _l = C.c_void_p, C.c_int, C.c_size_t, C.c_long
_fp = C.c_void_p  # device or host float*, passed as an address
SIGNATURES = {
    "jirf_abi_version": (_i, []),
    "jirf_device_count": (_i, []),
    "jirf_last_error": (C.c_char_p, []),
    "jirf_net_create": (_vp, [C.c_char_p, C.c_char_p, _i]),
    "jirf_net_destroy": (None, [_vp]),
    "jirf_net_set_param": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i]),
    "jirf_net_param_count": (_i, [_vp]),
    "jirf_net_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(_i)]),
    "jirf_net_get_param": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "jirf_net_finalize": (_i, [_vp]),
    "jirf_net_plan": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_sz)]),
    "jirf_net_forward": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_net_reset_cache": (_i, [_vp]),
    "jirf_net_get_cache": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i), _sz]),
    "jirf_net_set_cache": (_i, [_vp, _vp, _i, _i]),
    "jirf_net_attention_blocks": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), _i]),
    "jirf_net_get_perms": (_i, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "jirf_net_set_forced_perms": (_i, [_vp, _vp, _sz]),
    "jirf_net_enable_kernel_timing": (_i, [_vp, _i]),
    "jirf_net_kernel_times": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(_i), _i, C.POINTER(_i)]),
    "jirf_op_conv1x1": (_i, [_fp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_ln_conv1x1": (_i, [_fp, _fp, _fp, _i, _vp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_gdfn": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "jirf_op_ln_conv1x1_static": (_i, [_fp, _fp, _fp, _i, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_train_scratch_generation": (_l, []),
    "jirf_train_kernel_timing": (_i, [_i]),
    "jirf_train_kernel_times": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(_i), _i, C.POINTER(_i)]),
    "jirf_train_smooth_l1": (_i, [_fp, _fp, _fp, C.c_long, C.c_float, _fp, C.POINTER(C.c_double), _vp]),
    "jirf_train_sumsq": (_i, [_fp, C.c_long, _vp, C.POINTER(C.c_double), _vp]),
    "jirf_train_adam_step": (_i, [_fp, _fp, _fp, _fp, C.c_long, _vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float, _i, _i, _vp]),
    "jirf_train_conv": (_i, [_fp, _l, _fp, _i, _fp, _fp, _l, _fp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_conv_bwd_weight": (_i, [_fp, _l, _fp, _l, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_linear_bwd_weight": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_layernorm": (_i, [_fp, _l, _fp, _fp, _fp, _l, _i, _i, _i, _i, _vp]),
    "jirf_train_layernorm_bwd": (_i, [_fp, _l, _fp, _l, _fp, _fp, _l, _fp, _l, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_gelu": (_i, [_fp, _l, _fp, _l, _fp, _l, _i, _l, _i, _vp]),
    "jirf_train_gelu_bwd": (_i, [_fp, _l, _fp, _l, _fp, _l, _fp, _l, _fp, _l, _i, _l, _i, _vp]),
    "jirf_train_add": (_i, [_fp, _l, _fp, _l, _fp, _l, _i, _l, _vp]),
    "jirf_train_scale_add": (_i, [_fp, _l, _fp, _l, _fp, _fp, _l, _i, _l, _vp]),
    "jirf_train_channel_sum": (_i, [_fp, _l, _fp, _i, _i, _l, _i, _vp]),
    "jirf_train_drop_path_scales": (_i, [_fp, _fp, _i, _i, C.c_ulonglong, _fp, _vp]),
    "jirf_train_packed_floats": (_l, [_i, _i, _i, _i]),
    "jirf_train_pack_plan_create": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i),
                                        C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "jirf_train_pack_plan_run": (_i, [_vp, _vp]),
    "jirf_train_pack_plan_destroy": (None, [_vp]),
    "jirf_train_dwconv3x3": (_i, [_fp, _l, _fp, _fp, _fp, _l, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_dwconv3x3_bwd_weight": (_i, [_fp, _l, _fp, _l, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_window_attention": (_i, [_fp, _l, _fp, _fp, _fp, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_train_window_attention_bwd": (_i, [_fp, _l, _fp, _fp, _fp, _l, _fp, _l, _fp, _fp, _i, _i, _i, _i, _i, _i, _i,
                                             _i, _vp]),
    "jirf_op_gelu_error_scan": (_i, [C.c_float, C.POINTER(C.c_double), _vp]),
    "jirf_set_gemm_split_bf16": (_i, [_i]),
    "jirf_op_conv3x3": (_i, [_fp, _vp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_dwconv3x3": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_layernorm": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_reflect_pad": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_split_tiles": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "jirf_op_merge_tiles": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "jirf_op_crop_clamp": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_crop_clamp_u8": (_i, [_fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_psnr_work_doubles": (_i, []),
    "jirf_op_psnr_stats": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "jirf_op_ssim_work_doubles": (_i, [_i, _i, _i]),
    "jirf_op_ssim_stats": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, C.c_float, _vp, _vp]),
    "jirf_op_hint_attention_core": (_i, [_fp, _vp, _fp, _i, _fp, _fp, _vp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_head_attention_core": (_i, [_fp, _vp, _vp, _vp, C.c_float, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_cross_window_attention": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_dynamic_filter": (_i, [_fp, _vp, _i, _fp, _fp, _i, _i, _i, _i, _vp]),
    "jirf_op_resize": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "jirf_op_fspg_high": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
}


def lib() -> C.CDLL:
    """Loads the shared library once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C image-restoration_amd` "
                "(python -c 'import __graft_entry__ as g; g.build()').")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    return (lib().jirf_last_error() or b"").decode()


def check(rc: int) -> None:
    if rc != 0:
        raise JirfError(rc, last_error())


def _host_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class Net:
    """One network instance bound to one GPU (jirf_net*)."""

    def __init__(self, arch: str, cfg: dict, device: int = 0):
        L = lib()
        cfg_json = json.dumps({k: v for k, v in cfg.items()})
        self._h = L.jirf_net_create(arch.encode(), cfg_json.encode(), int(device))
        if not self._h:
            msg = last_error()
            # mirror the reference's constructor/registry failures
            if "is not found" in msg:
                raise ValueError(msg)
            if "unexpected keyword argument" in msg:
                raise TypeError(msg)
            raise JirfError(-1, msg)
        self.arch = arch
        self.device = device
        self.finalized = False

    def close(self):
        if getattr(self, "_h", None):
            lib().jirf_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ------------------------------------------------------------------------------
    def param_specs(self) -> List[Tuple[str, Tuple[int, ...]]]:
        L = lib()
        out = []
        key = C.c_char_p()
        shape = (C.c_int64 * 8)()
        nd = C.c_int()
        for i in range(L.jirf_net_param_count(self._h)):
            check(L.jirf_net_param_info(self._h, i, C.byref(key), shape, C.byref(nd)))
            out.append((key.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        return out

    def set_param(self, key: str, value) -> None:
        a = np.ascontiguousarray(np.asarray(value))
        if a.dtype not in _DTYPES:
            a = a.astype(np.float32)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        check(lib().jirf_net_set_param(self._h, key.encode(), _host_ptr(a), _DTYPES[a.dtype], shape, a.ndim))

    def get_param(self, key: str, shape: Sequence[int]) -> np.ndarray:
        out = np.empty(tuple(shape), dtype=np.float32)
        check(lib().jirf_net_get_param(self._h, key.encode(), _host_ptr(out), out.size))
        return out

    def finalize(self) -> None:
        check(lib().jirf_net_finalize(self._h))
        self.finalized = True

    # ---- execution -------------------------------------------------------------------------------
    def plan(self, b: int, h: int, w: int, mode: int = BATCH_PER_IMAGE) -> int:
        n = C.c_size_t()
        check(lib().jirf_net_plan(self._h, b, h, w, mode, C.byref(n)))
        return n.value

    def forward_ptr(self, x_ptr: int, y_ptr: int, b: int, h: int, w: int, mode: int, stream: int = 0) -> None:
        check(lib().jirf_net_forward(self._h, x_ptr, y_ptr, b, h, w, mode, 1, stream))

    def forward_host(self, x: np.ndarray, mode: int = BATCH_PER_IMAGE, out_channels: Optional[int] = None) -> np.ndarray:
        x = _f32(x)
        b, _, h, w = x.shape
        y = np.empty((b, out_channels or x.shape[1], h, w), dtype=np.float32)
        check(lib().jirf_net_forward(self._h, _host_ptr(x), _host_ptr(y), b, h, w, mode, 0, None))
        return y

    # ---- HINT state / diagnostics ---------------------------------------------------------------
    def reset_cache(self) -> None:
        check(lib().jirf_net_reset_cache(self._h))

    def get_cache(self, max_c: int = 1024, max_n: int = 64) -> Optional[np.ndarray]:
        buf = np.empty(max_n * max_c * max_c // 8, dtype=np.float32)
        n, c = C.c_int(), C.c_int()
        rc = lib().jirf_net_get_cache(self._h, _host_ptr(buf), C.byref(n), C.byref(c), buf.size)
        if rc == -3:  # JIRF_ESTATE: None
            return None
        check(rc)
        return buf[: n.value * c.value * c.value].reshape(n.value, c.value, c.value).copy()

    def set_cache(self, cache: Optional[np.ndarray]) -> None:
        if cache is None:
            self.reset_cache()
            return
        a = _f32(cache)
        check(lib().jirf_net_set_cache(self._h, _host_ptr(a), a.shape[0], a.shape[-1]))

    def attention_blocks(self) -> List[int]:
        n = C.c_int()
        chans = (C.c_int * 256)()
        check(lib().jirf_net_attention_blocks(self._h, C.byref(n), chans, 256))
        return [int(chans[i]) for i in range(n.value)]

    def get_perms(self, groups: int) -> List[np.ndarray]:
        """Channel orders used by the last forward: one [groups, C] int32 array per attention block."""
        chans = self.attention_blocks()
        total = sum(chans) * groups
        buf = np.empty(total, dtype=np.int32)
        n = C.c_size_t()
        check(lib().jirf_net_get_perms(self._h, _host_ptr(buf), buf.size, C.byref(n)))
        out, off = [], 0
        for c in chans:
            out.append(buf[off: off + groups * c].reshape(groups, c).copy())
            off += groups * c
        return out

    def set_forced_perms(self, perms: Optional[List[np.ndarray]]) -> None:
        if perms is None:
            check(lib().jirf_net_set_forced_perms(self._h, None, 0))
            return
        flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int32).reshape(-1) for p in perms]))
        check(lib().jirf_net_set_forced_perms(self._h, _host_ptr(flat), flat.size))

    def enable_kernel_timing(self, on: bool) -> None:
        check(lib().jirf_net_enable_kernel_timing(self._h, int(on)))

    def kernel_times(self) -> List[dict]:
        cap = 512
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        ln = (C.c_int * cap)()
        n = C.c_int()
        check(lib().jirf_net_kernel_times(self._h, names, ms, fl, by, ln, cap, C.byref(n)))
        return [dict(name=names[i].decode(), ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i]),
                     launches=int(ln[i])) for i in range(min(n.value, cap))]
