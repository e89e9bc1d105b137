"""Phase timeline of the activation-stationary GEMM from the stamps of the diagnostic library
(profiles/exp/build_probe.sh).  Usage, on the GPU box from the repository root:
    JIRF_HIP_LIB=profiles/exp/libjirf_probe.so python3 profiles/exp/xs_probe.py [B Cin Cout P]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "image-restoration_amd"))
import jirf_hip  # noqa: E402

B, C, M, P = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 96, 510, 65536)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, C, P), dtype=np.float32)).cuda()
lw, lb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
w = np.ascontiguousarray(rng.standard_normal((M, C), dtype=np.float32) * 0.1)
y = torch.empty(B, M, P, device="cuda")
vp = ctypes.c_void_p
lib = jirf_hip.lib()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(6):
    if i == 5:
        ev[0].record()
    rc = lib.jirf_op_ln_conv1x1(vp(x.data_ptr()), vp(lw.data_ptr()), vp(lb.data_ptr()), 1, w.ctypes.data_as(vp),
                                vp(y.data_ptr()), B, C, M, P, None)
    assert rc == 0, jirf_hip.last_error()
ev[1].record()
torch.cuda.synchronize()
print(f"last call (incl. weight upload + sync): {ev[0].elapsed_time(ev[1]):.3f} ms")
SL, WG = 48, 8192
buf = np.zeros(WG * 4 * SL, np.uint64)
raw = ctypes.CDLL(jirf_hip.LIB_PATH)
assert raw.jirf_debug_xs_probe(buf.ctypes.data_as(vp)) == 0
st = buf.reshape(WG, 4, SL).astype(np.int64)
n_wg = min(WG, int((st[:, 0, 0] != 0).sum()))
st = st[:n_wg]
t0 = st[:, :, 0]
hw = st[:, 0, 1]
xcc, hwid = hw >> 32, hw & 0xffffffff
cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
print(f"{n_wg} workgroups on {len(np.unique(cu))} distinct (xcc, se, sh, cu)")
n_items = int(((st[0, 0, 4:40:2] != 0)).sum())
us = lambda c: c / 2390.0
print("median per-wave phase lengths (us at 2.39 GHz):")
print(f"  tile load (entry -> barrier)      {us(np.median(st[:, :, 2] - t0)):8.2f}")
print(f"  LayerNorm (stats + normalise)     {us(np.median(st[:, :, 3] - st[:, :, 2])):8.2f}")
if st[0, 0, 40]:
    seq = [(0, "entry"), (40, "tile loads issued"), (2, "tile arrived, local sums"), (42, "mean exchanged"),
           (43, "variance exchanged"), (44, "normalised tile written"), (3, "barrier passed")]
    for (a_, _), (b__, name) in zip(seq[:-1], seq[1:]):
        print(f"    -> {name:28s} {us(np.median(st[:, :, b__] - st[:, :, a_])):8.2f}")
prev = st[:, :, 3]
for i in range(n_items):
    a, b_ = st[:, :, 4 + 2 * i], st[:, :, 5 + 2 * i]
    print(f"  item {i}: MFMA loop {us(np.median(a - prev)):8.2f}   epilogue issue {us(np.median(b_ - a)):8.2f}")
    prev = b_
life = st[:, :, 47] - t0
print(f"  wave lifetime                      {us(np.median(life)):8.2f}  (p10 {us(np.percentile(life, 10)):.2f}, p90 {us(np.percentile(life, 90)):.2f})")
# co-residency: per CU, how many workgroups overlap a workgroup's life on average
order = np.argsort(cu, kind="stable")
res = []
for c in np.unique(cu)[:64]:
    idx = np.where(cu == c)[0]
    s, e = st[idx, 0, 0], st[idx, 0, 47]
    for j in range(len(idx)):
        ov = np.minimum(e, e[j]) - np.maximum(s, s[j])
        res.append(np.clip(ov, 0, None).sum() / max(e[j] - s[j], 1))
print(f"  mean workgroups resident per CU while one runs: {np.mean(res):.2f}")
span = [(st[cu == c, 0, 47].max() - st[cu == c, 0, 0].min()) for c in np.unique(cu)]
print(f"  per-CU span first entry -> last exit: median {us(np.median(span)):.1f} us over {np.median([np.sum(cu == c) for c in np.unique(cu)]):.0f} workgroups")

# per SIMD: fraction of the span during which at least one wave is inside an MFMA loop, and the mean number
hw_w = st[:, :, 1]
simd = (hw_w & 0xffffffff) >> 4 & 3
cov, dens = [], []
for c in np.unique(cu)[:48]:
    for sd in range(4):
        sel = (cu[:, None] == c) & (simd == sd)
        if not sel.any():
            continue
        ev = []
        for i in range(n_items):
            start = st[:, :, 3] if i == 0 else st[:, :, 5 + 2 * (i - 1)]
            ev += [(t, 1) for t in start[sel]] + [(t, -1) for t in st[:, :, 4 + 2 * i][sel]]
        ev.sort()
        t_lo, t_hi = st[:, :, 0][sel].min(), st[:, :, 47][sel].max()
        depth, last, busy, area = 0, t_lo, 0, 0
        for t, d in ev:
            if depth > 0:
                busy += t - last
            area += depth * (t - last)
            depth += d
            last = t
        cov.append(busy / (t_hi - t_lo))
        dens.append(area / (t_hi - t_lo))
print(f"  per SIMD: some wave inside an MFMA loop {100 * np.mean(cov):.1f} % of the time; mean waves inside a loop {np.mean(dens):.2f}")
