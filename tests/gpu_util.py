"""Helpers for the GPU parity tests: torch is used only to own device memory."""
import ctypes as C

import numpy as np
import torch

import jirf_hip


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def hptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    jirf_hip.check(getattr(jirf_hip.lib(), name)(*args))
    torch.cuda.synchronize()


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
