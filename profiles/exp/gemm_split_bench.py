#!/usr/bin/env python3
"""Device time and accuracy of the LayerNorm + 1x1 GEMM entry point with the fp32-MFMA and the split-bf16 kernels.
The kernel choice is read once per process, so each setting runs in a child:  python profiles/exp/gemm_split_bench.py"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("ln", 32, 96, 510, 65536), ("ln", 32, 96, 288, 65536), ("ln", 32, 96, 510, 16384), ("ln", 32, 48, 254, 65536),
          ("ln", 1, 96, 510, 65536), ("ln", 32, 192, 1020, 4096), ("ln", 32, 192, 576, 4096), ("ln", 32, 384, 2042, 1024),
          ("ln", 32, 384, 1152, 1024), ("res", 32, 510, 192, 4096), ("res", 32, 1021, 384, 1024), ("res", 32, 384, 192, 4096),
          ("res", 32, 768, 384, 1024), ("res", 32, 192, 192, 4096), ("res", 32, 384, 384, 1024),
          ("res", 32, 255, 96, 65536), ("res", 32, 192, 96, 65536), ("res", 32, 255, 96, 16384), ("res", 32, 127, 48, 65536),
          ("res", 32, 96, 96, 65536), ("res", 1, 255, 96, 65536)]
if os.environ.get("GEMM_SPLIT_IDX"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["GEMM_SPLIT_IDX"].split(",")]
if os.environ.get("GEMM_SPLIT_SHAPES"):
    SHAPES = [t for t in SHAPES if t[0] in os.environ["GEMM_SPLIT_SHAPES"].split(",")]


def child():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "image-restoration_amd")]
    import numpy as np
    import torch
    import jirf_hip
    L = jirf_hip.lib()
    rng = np.random.default_rng(0)
    hp = lambda a: a.ctypes.data_as(C.c_void_p)

    def times():
        cap = 64
        names = (C.c_char_p * cap)(); ms = (C.c_float * cap)(); fl = (C.c_double * cap)(); by = (C.c_double * cap)()
        ln = (C.c_int * cap)(); n = C.c_int()
        jirf_hip.check(L.jirf_train_kernel_times(names, ms, fl, by, ln, cap, C.byref(n)))
        return [(names[i].decode(), ms[i], fl[i], by[i], ln[i]) for i in range(n.value)]

    out = {}
    for (kind, B, Cin, Cout, P) in SHAPES:
        x = torch.randn(B, Cin, P, device="cuda") * 2 + 0.5
        y = torch.empty(B, Cout, P, device="cuda")
        w = np.ascontiguousarray((rng.standard_normal((Cout, Cin)) * Cin ** -0.5).astype(np.float32))
        lw = np.ascontiguousarray((1 + 0.1 * rng.standard_normal(Cin)).astype(np.float32))
        lb = np.ascontiguousarray((0.1 * rng.standard_normal(Cin)).astype(np.float32))
        lwd, lbd = torch.from_numpy(lw).cuda(), torch.from_numpy(lb).cuda()
        res = torch.randn(B, Cout, P, device="cuda") if kind == "res" else None
        if kind == "ln":
            run = lambda: jirf_hip.check(L.jirf_op_ln_conv1x1(C.c_void_p(x.data_ptr()), C.c_void_p(lwd.data_ptr()), C.c_void_p(lbd.data_ptr()), 1,
                                                              hp(w), C.c_void_p(y.data_ptr()), B, Cin, Cout, P, None))
        else:
            run = lambda: jirf_hip.check(L.jirf_op_conv1x1(C.c_void_p(x.data_ptr()), hp(w), None, C.c_void_p(res.data_ptr()),
                                                           C.c_void_p(y.data_ptr()), B, Cin, Cout, P, None))
        for _ in range(2):
            run()
        jirf_hip.check(L.jirf_train_kernel_timing(1))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = [r for r in times() if r[0].startswith("gemm")]
        jirf_hip.check(L.jirf_train_kernel_timing(0))
        name, ms, fl, by, n = t[0]
        # accuracy on the first image's first 4096 pixels against float64
        npx = min(P, 4096)
        xs = x[0, :, :npx].double().cpu().numpy()
        if kind == "ln":
            mu = xs.mean(0, keepdims=True)
            nrm = (xs - mu) / np.sqrt(xs.var(0, keepdims=True) + 1e-5) * lw[:, None] + lb[:, None]
            ref = w.astype(np.float64) @ nrm
        else:
            ref = w.astype(np.float64) @ xs + res[0, :, :npx].double().cpu().numpy()
        err = float(np.abs(y[0, :, :npx].double().cpu().numpy() - ref).max())
        out[f"{kind} {B}x{Cin}->{Cout}x{P}"] = dict(us=ms / n * 1e3, tflops=fl / (ms / 1e3) / 1e12, gbs=by / (ms / 1e3) / 1e9, err=err, label=name)
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("GEMM_SPLIT_CHILD"):
        child()
    else:
        res = {}
        for split in ("0", "1"):
            env = dict(os.environ, GEMM_SPLIT_CHILD="1", JIRF_GEMM_SPLIT_BF16=split, JIRF_OP_LN_FOLD="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print(r.stdout[-2000:], r.stderr[-3000:])
                sys.exit(1)
            res[split] = json.loads(r.stdout.strip().splitlines()[-1])
        for k in res["0"]:
            a, b = res["0"][k], res["1"][k]
            print(f"{k:24s} fp32 {a['us']:8.1f} us {a['tflops']:6.1f} TF {a['gbs']:7.0f} GB/s err {a['err']:.2e} | "
                  f"split {b['us']:8.1f} us {b['tflops']:6.1f} TF {b['gbs']:7.0f} GB/s err {b['err']:.2e}")
