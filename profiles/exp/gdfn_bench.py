#!/usr/bin/env python3
"""Device time of the gated feed-forward block as one kernel and as three launches (jirf_op_gdfn), per launch label,
from the library's HIP-event clock.  python profiles/exp/gdfn_bench.py [B H W reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-restoration_amd")]
import numpy as np
import torch

import jirf_hip

FUSED_ONLY = bool(int(os.environ.get("GDFN_FUSED_ONLY", "0")))
B, H, W, reps = (int(v) for v in (sys.argv[1:5] + ["32", "256", "256", "5"][len(sys.argv) - 1:]))
Cn, h = 96, 255
L = jirf_hip.lib()
rng = np.random.default_rng(0)
f = lambda *s, scale=1.0: np.ascontiguousarray((rng.standard_normal(s) * scale).astype(np.float32))
lw, lb, win, wdw, wout = 1 + 0.1 * f(Cn), 0.1 * f(Cn), f(2 * h, Cn, scale=Cn ** -0.5), f(2 * h, 9, scale=0.4), f(Cn, h, scale=h ** -0.5)
lw = np.ascontiguousarray(lw.astype(np.float32)); lb = np.ascontiguousarray(lb.astype(np.float32))
x = torch.randn(B, Cn, H, W, device="cuda")
y = torch.empty_like(x)
hp = lambda a: a.ctypes.data_as(C.c_void_p)


def times():
    cap = 64
    names = (C.c_char_p * cap)(); ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); by = (C.c_double * cap)()
    ln = (C.c_int * cap)(); n = C.c_int()
    jirf_hip.check(L.jirf_train_kernel_times(names, ms, fl, by, ln, cap, C.byref(n)))
    return [(names[i].decode(), ms[i], fl[i], by[i], ln[i]) for i in range(n.value)]


for fused in ((1,) if FUSED_ONLY else (1, 0)):
    for _ in range(2):
        jirf_hip.check(L.jirf_op_gdfn(x.data_ptr(), y.data_ptr(), B, Cn, h, H, W, 1, hp(lw), hp(lb), hp(win), None, hp(wdw), None,
                                      hp(wout), None, fused, None))
    jirf_hip.check(L.jirf_train_kernel_timing(1))
    for _ in range(reps):
        jirf_hip.check(L.jirf_op_gdfn(x.data_ptr(), y.data_ptr(), B, Cn, h, H, W, 1, hp(lw), hp(lb), hp(win), None, hp(wdw), None,
                                      hp(wout), None, fused, None))
    tot = 0.0
    for name, ms, fl, by, n in times():
        print(f"{'one kernel' if fused else 'three launches':15s} {ms / n * 1e3:9.1f} us/launch  {fl / (ms / 1e3) / 1e12 if ms else 0:7.2f} TFLOP/s  "
              f"{by / (ms / 1e3) / 1e9 if ms else 0:8.1f} GB/s  {name}")
        tot += ms / n
    print(f"{'one kernel' if fused else 'three launches':15s} {tot * 1e3:9.1f} us total per block  (B={B} {H}x{W})")
    jirf_hip.check(L.jirf_train_kernel_timing(0))
