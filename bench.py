#!/usr/bin/env python3
"""Throughput of the HINT inference forward on MI355X (BASELINE.json configs[1]).

A step = one forward of libjirf_hip's HINT over one batch of synthetic 256x256 RGB crops with
sigma=50 Gaussian noise, fp32, inputs and outputs resident in HBM.  Every image is an
independent cold-start call (the reference's test scripts run B=1; `per_image` batch mode is
bit-for-bit B separate calls), so ranks shard by image and no collective touches the data path.

Prints ONE JSON line on rank 0 (see --help for the knobs).  Extra legs, N=1 only:
  roofline      the dominant kernel (the fp32-MFMA pointwise-conv GEMM family) timed with HIP
                events on the launch stream in a second pass over the same steps
  cpu_baseline  the CPU oracle (a torch-CPU restatement of the reference forward; the reference's
                Jittor runtime is not installable here) on a bounded sample, host cores stated
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-restoration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X fp32 matrix peak (v_mfma_f32_32x32x2_f32), dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X bf16 matrix peak (v_mfma_f32_32x32x16_bf16), dense
SPLIT_TERMS = 6                # bf16 partial products the split GEMM kernels issue per fp32 product
HBM_PEAK_GBS = 8000.0


def hint_flops_per_image(cfg, h, w):
    """2*MAC count of one HINT forward (constructor shapes of hint_arch.py of the reference)."""
    d, nb, ffn = cfg["dim"], cfg["num_blocks"], cfg["ffn_expansion_factor"]
    p1 = h * w
    total = 0.0

    def block(c, p, att):
        hid = int(c * ffn)
        f = 2 * c * 2 * hid * p + 18 * 2 * hid * p + 2 * hid * c * p
        if att:
            f += 2 * c * 3 * c * p + 18 * 3 * c * p + 2 * c * c * p          # qkv, dw, project_out
            f += 2 * c * c * p                                              # corrcoef
            gs = [int(r / 8 * c) for r in (1, 2, 2, 3)]
            f += sum(4 * g * g * p for g in gs)                            # QK^T + AV
            f += 2 * c * c * p + 2 * c * (c // 2) * p * 2                   # intra modulator
        return f

    total += 2 * 27 * d * p1
    plan = [(nb[0], d, p1, False), (nb[1], 2 * d, p1 // 4, False), (nb[2], 4 * d, p1 // 16, False),
            (nb[1], 8 * d, p1 // 64, True), (nb[2], 4 * d, p1 // 16, True), (nb[1], 2 * d, p1 // 4, True),
            (nb[0], 2 * d, p1, True), (cfg["num_refinement_blocks"], 2 * d, p1, True)]
    for n, c, p, att in plan:
        total += n * block(c, p, att)
    total += 2 * 9 * d * (d // 2) * p1 + 2 * 9 * 2 * d * d * (p1 // 4) + 2 * 9 * 4 * d * 2 * d * (p1 // 16)
    total += 2 * 9 * 8 * d * 16 * d * (p1 // 64) + 2 * 9 * 4 * d * 8 * d * (p1 // 16) + 2 * 9 * 2 * d * 4 * d * (p1 // 4)
    total += 2 * 8 * d * 4 * d * (p1 // 16) + 2 * 4 * d * 2 * d * (p1 // 4)
    total += 2 * 9 * 2 * d * 3 * p1
    return total


def kernel_source_digest():
    """sha256 over the GEMM kernel sources and the argument structure they are launched with: the PMC traffic
    record under profiles/ is only valid for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "image-restoration_amd", "csrc")
    for f in ("gemm.hip", "gemm_split.hip", "split_bf16.h"):
        with open(os.path.join(src, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(src, "common.h")) as fh:
        text = fh.read()
    h.update(text[text.index("struct GemmArgs {"):text.index("void launch_gemm(")].encode())
    return h.hexdigest()


def pmc_traffic(label, batch):
    """HBM bytes per launch of the kernel `label`, from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), with the
    kernel symbol and grid it was measured on.  (None, None) when no pass was recorded for this launch
    shape and batch, or when the kernel sources have changed since the pass (stale record)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        with open(path) as fh:
            rec = json.load(fh)
        if not (rec.get("op", "").startswith(label + " ") and f"(B={batch})" in rec.get("op", "")):
            continue
        if rec.get("kernel_source_sha256") != kernel_source_digest():
            return None, {"stale_record": os.path.basename(path)}
        for sym, k in rec.get("kernels", {}).items():
            return k.get("hbm_bytes_per_launch"), {"record": os.path.basename(path), "kernel_symbol": sym,
                                                   "grid_x": k.get("grid_x")}
    return None, None


def _dominant(detail):
    """(name, record) of the launch shape with the largest share of device time, with its roofline fraction."""
    name, a = max(detail.items(), key=lambda kv: kv[1]["ms"])
    family = name.split(" ")[0]
    per_launch_s = a["ms"] / 1e3 / a["launches"]
    if family.endswith("_bf16x3"):
        # split GEMM: the matrix pipe executes SPLIT_TERMS bf16 products per algorithmic fp32 product; the bound is
        # whichever of the two floors (HBM bytes at 8 TB/s, issued bf16 flops at the dense bf16 peak) is the longer
        issued = SPLIT_TERMS * a["flops"] / a["launches"]
        nbytes = a["bytes"] / a["launches"]
        if nbytes / (HBM_PEAK_GBS * 1e9) >= issued / (MFMA_BF16_PEAK_TFLOPS * 1e12):
            ach = nbytes / per_launch_s / 1e9
            return name, {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                          "issued_bf16_tflops": round(issued / per_launch_s / 1e12, 1)}
        ach = issued / per_launch_s / 1e12
        return name, {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                      "note": f"issued bf16 flops = {SPLIT_TERMS} x the algorithmic fp32 flops"}
    if family.startswith("gemm") or family == "hint_chan_stats" or family.startswith("win_attn") or \
            family in ("train_dw", "train_win_attn_bwd"):
        ach = a["flops"] / a["launches"] / per_launch_s / 1e12
        return name, {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4)}
    ach = a["bytes"] / a["launches"] / per_launch_s / 1e9
    return name, {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}


def run_extra_config(nets, arch, B, Hh, Ww, steps, warmup=2, crop=0, overlap=0, seed=7, carry=False):
    """One more configuration of BASELINE.json on this GPU (N = 1 only): the same timing discipline as the
    headline (warm-up, K steps between synchronisations, inputs resident in HBM), then one pass with HIP
    events around every launch for the model FLOP rate and the dominant launch shape."""
    from jirf_hip import harness
    from jirf_hip.synthetic import default_config, make_state, synthetic_images
    from basicsr.models.archs import define_network
    if arch not in nets:
        cfg = default_config(arch)
        net = define_network(dict(cfg, type=arch))
        net.load_state_dict(make_state(arch, cfg, seed=1))
        net.cuda(torch.cuda.current_device()).eval()
        net.batch_mode = "per_image"
        nets[arch] = net
    net = nets[arch]
    x = torch.from_numpy(synthetic_images(B, Hh, Ww, seed=seed, sigma=50 if arch == "HINT" else 0)).cuda()
    need = {"HINT": 8, "ASTv2": 8, "FPro": 32, "AST": 128}[arch]
    script = Hh % need != 0 or Ww % need != 0 or (arch == "AST" and Hh != Ww) or crop > 0

    def step():
        if arch == "HINT":
            net.qv_cache = None
        if crop:
            return harness.restore_image(net, x, factor=8, crop_size=crop, overlap_size=overlap, carry_cache=carry)
        if script:
            return harness.restore_image(net, x, factor=need, square_multiple=128 if arch == "AST" else None)
        return net(x)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert bool(torch.isfinite(y).all()), "non-finite output"
    if carry:
        net.qv_cache = None
        net.batch_mode = "per_image"
    nat = net.native()
    nat.enable_kernel_timing(True)
    detail, flops = {}, 0.0
    step()
    torch.cuda.synchronize()
    for k in nat.kernel_times():
        a = detail.setdefault(k["name"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        for f in ("ms", "flops", "bytes", "launches"):
            a[f] += k[f]
        flops += k["flops"]
    nat.enable_kernel_timing(False)
    rec = {"arch": arch, "batch": B, "height": Hh, "width": Ww,
           "path": (f"script loop body: reflect pad to x8, {crop}x{crop} tiles overlap {overlap}"
                    + (", one B = 1 call per tile in list order with qv_cache carried from tile to tile "
                       "(Motion_Deblurring/test_GoPro.py:142-171)" if carry else "") + ", mean blend, crop, clamp" if crop else
                    ("script loop body: reflect pad" + (" to the enclosing square (side % 128 == 0)" if arch == "AST" else f" to x{need}")
                     + ", forward, crop, clamp") if script else "forward"),
           "steps": steps, "ms_per_step": round(ms, 3), "ms_per_image": round(ms / B, 3),
           "mp_per_s": round(B * Hh * Ww / 1e6 / (ms / 1e3), 3),
           "model_tflops_per_s": round(flops / (ms / 1e3) / 1e12, 2),
           "frac_of_f32_mfma_peak": round(flops / (ms / 1e3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
    if detail:
        rec["dominant_kernel"] = _dominant(detail)[1]
    return rec


def train_step_bench(args, rank, local_rank, world, use_dist):
    """One AST optimisation step per "step" (reference basicsr/models/image_restoration_model.py:235-276: forward,
    smooth-L1, backward, gradient mean over the ranks, clip 0.01, Adam) on a batch of --train-batch crops per GPU
    (Enhancement/Options/AST_LOL_v2_synthetic.yml: mini-batch 2 at the 256-pixel stage of the progressive schedule).
    Every rank holds a full replica and its own crops; the only exchange is the all-reduce of the flat gradient
    buffer (RCCL), plus the scalar mean of the logged loss."""
    import torch.distributed as dist
    from jirf_hip.synthetic import default_config, make_state, synthetic_images
    from jirf_hip.train import AstTrainStep
    cfg = default_config("AST")
    sd = make_state("AST", cfg, seed=1)
    B, S = args.train_batch, args.size
    # stochastic depth at the reference's constructor default (drop_path_rate 0.1) unless --drop-path says otherwise;
    # each rank draws its own keep decisions, like the reference's per-process generator
    tr = AstTrainStep(cfg, sd, device=local_rank, lr=2e-4, use_grad_clip=True, coupled=True,
                      drop_path_rate=args.drop_path, seed=rank, use_graph=not args.no_graph)
    tr.broadcast_parameters()
    gt = torch.from_numpy(synthetic_images(B, S, S, seed=500 + rank, sigma=0)).cuda()
    lq = torch.from_numpy(synthetic_images(B, S, S, seed=500 + rank, sigma=25)).cuda()

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the loss of every step stays on the device until the run is over (the reference's loss is a lazy Jittor
    # variable, evaluated when the logger prints it): nothing in the loop waits for the GPU
    losses_dev = []
    for _ in range(args.warmup):
        losses_dev.append(tr.optimize_parameters(lq, gt, sync_loss=False).clone())
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses_dev.append(tr.optimize_parameters(lq, gt, sync_loss=False).clone())
    host_elapsed = time.perf_counter() - t0               # the host's share: queueing the steps
    sync_all()
    elapsed = time.perf_counter() - t0
    losses = [float(t.item()) for t in losses_dev]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(np.isfinite(losses)), "non-finite loss"
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": f"trained megapixels/sec (node), AST optimisation step, {S}x{S} crops",
        "value": round(world * B * S * S / 1e6 / (ms_per_step / 1e3), 3), "unit": "MP/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "images_per_s": round(world * B / (ms_per_step / 1e3), 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (uniform [0,1) RGB targets, inputs = targets + N(0,(25/255)^2); seeded random weights in the "
                "reference key layout, broadcast from rank 0)",
        "config": {"workload": f"AST (reference default constructor, token_mlp frfn, {tr.num_parameters()} parameters) "
                               f"optimisation step on {B} {S}x{S} crops per GPU: forward, smooth-L1, backward, gradient "
                               f"all-reduce (mean), clip 0.01, Adam; fp32; stochastic depth (DropPath) rate {tr.drop_path_rate:g}",
                   "batch_per_gpu": B, "height": S, "width": S,
                   "sharding": "one replica per rank, one all-reduce of the flat gradient buffer per step",
                   "gradient_floats": tr.n_flat},
        "loss_first": round(losses[0], 6), "loss_last": round(losses[-1], 6),
        "host_queueing_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
    }
    if use_dist:
        # the exchange on its own: the same all-reduce of the flat gradient buffer, timed outside the step (with one
        # rank under the launcher RCCL still runs the collective on the device)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(5):
            tr.average_gradients()
        sync_all()
        out["allreduce_ms"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
        out["allreduce_floats"] = tr.n_flat
    if rank == 0 and world == 1 and not args.no_roofline:
        tr.kernel_timing(True)
        acc, detail = {}, {}
        for _ in range(max(1, min(args.steps, 3))):
            tr.optimize_parameters(lq, gt)
            torch.cuda.synchronize()
            for k in tr.kernel_times():
                for table, key in ((acc, k["name"].split(" ")[0]), (detail, k["name"])):
                    a = table.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                    for f in ("ms", "flops", "bytes", "launches"):
                        a[f] += k[f]
        tr.kernel_timing(False)
        n_rep = max(1, min(args.steps, 3))
        if args.detail:
            with open(args.detail, "w") as fh:
                for k, v in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
                    fh.write(f"{v['ms'] / n_rep:9.3f} ms/step  {v['launches'] // n_rep:4d} launches  "
                             f"{v['flops'] / (v['ms'] / 1e3) / 1e12 if v['ms'] else 0:7.2f} TFLOP/s  "
                             f"{v['bytes'] / (v['ms'] / 1e3) / 1e9 if v['ms'] else 0:8.1f} GB/s  {k}\n")
        tot_ms = sum(a["ms"] for a in acc.values())
        name, roof = _dominant(detail)
        a = detail[name]
        roof.update({"traffic": None, "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                     "algorithmic_flops_per_launch": a["flops"] / a["launches"],
                     "launches_per_step": a["launches"] // n_rep,
                     "avg_launch_us": round(a["ms"] / a["launches"] * 1e3, 2),
                     "share_of_device_time": round(a["ms"] / tot_ms, 4)})
        out["roofline"] = roof
        out["device_ms_per_step"] = round(tot_ms / n_rep, 3)
        out["kernels"] = {k: {"ms_per_step": round(v["ms"] / n_rep, 3), "launches_per_step": v["launches"] // n_rep,
                              "tflops": round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) if v["ms"] > 0 else 0,
                              "gbs": round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) if v["ms"] > 0 else 0}
                          for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: forward + backward of ONE crop of the batch through the oracle with torch autograd on the
        # host cores (the only place the benchmark touches oracle/)
        import torch.nn.functional as F
        from oracle.ast_oracle import ast_forward
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        torch.set_num_threads(cores)
        params = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).requires_grad_(True)
                  for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
        t0 = time.perf_counter()
        pred = ast_forward(params, lq[:1].cpu(), cfg)
        F.smooth_l1_loss(pred, gt[:1].cpu(), reduction="none").mean().backward()
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(S * S / 1e6 / dt, 5), "unit": "MP/s", "cores": cores, "kind": "port",
                               "sample": f"forward + backward of 1 crop {S}x{S} through the oracle/ restatement of AST with "
                                         f"torch autograd (torch-CPU fp32, {cores} threads), {dt:.1f} s; no optimiser update"}
    return out


def images_bench(args, rank, local_rank, world, use_dist):
    """BASELINE configs[2] / [3] as the scripts run them: a folder of --images equal-sized images restored by
    `harness.restore_images` - image i on rank i % N, each through the loop body of the test scripts (reflect pad,
    forward, crop, clamp), --batch of them in flight per call with per-image semantics, results gathered on rank 0.
    The folder is fixed, so the work per GPU falls as N grows ("strong" scaling).  A step = the whole folder."""
    import torch.distributed as dist
    from jirf_hip import harness
    from jirf_hip.synthetic import default_config, make_state, synthetic_images
    from basicsr.models.archs import define_network
    cfg = default_config(args.arch)
    net = define_network(dict(cfg, type=args.arch))
    net.load_state_dict(make_state(args.arch, cfg, seed=1))
    net.cuda(local_rank).eval()
    N, Hh, Ww = args.images, args.height or 321, args.width or 481
    # image i is the same whatever the number of ranks; a rank uploads only the ones it will restore
    mine = set(range(rank, N, world))
    imgs = [torch.from_numpy(synthetic_images(1, Hh, Ww, seed=1000 + i, sigma=50 if args.arch == "HINT" else 0)[0]).cuda()
            if i in mine else torch.empty(3, Hh, Ww) for i in range(N)]
    kw = dict(batch=args.batch)
    if args.crop_size:
        kw.update(factor=8, crop_size=args.crop_size, overlap_size=args.overlap_size)
    elif args.arch == "AST":
        kw.update(square_multiple=128)
    else:
        kw.update(factor=8)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        harness.restore_images(net, imgs, **kw)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = harness.restore_images(net, imgs, **kw)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        assert len(got) == N and all(tuple(g.shape) == (3, Hh, Ww) and bool(torch.isfinite(g).all()) for g in got)
    ms = elapsed / args.steps * 1e3
    return {"metric": f"restored megapixels/sec (node), {args.arch} over a folder of {N} {Ww}x{Hh} images",
            "value": round(N * Hh * Ww / 1e6 / (ms / 1e3), 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "ms_per_image": round(ms / N, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (uniform [0,1) RGB images, seeded per image index; seeded random weights in the reference key layout)",
            "config": {"workload": f"{args.arch} (reference default constructor) over {N} {Ww}x{Hh} RGB images, each through the "
                                   "test scripts' loop body (reflect pad, forward, crop, clamp), fp32; results gathered on rank 0 "
                                   "inside the timed region",
                       "images": N, "height": Hh, "width": Ww, "batch_in_flight": args.batch,
                       "sharding": "image i on rank i % N, weights replicated, no data-path collective; one gather of the results"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32,
                    help="256x256 crops per GPU per step (images are independent units: the batch only sets how many "
                         "are in flight; 16 -> 32 gains 3.5 %, 32 -> 64 nothing)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--height", type=int, default=0, help="image height (default: --size)")
    ap.add_argument("--width", type=int, default=0, help="image width (default: --size)")
    ap.add_argument("--arch", default="HINT", choices=["HINT", "ASTv2", "AST", "FPro"],
                    help="network; the headline workload is HINT (the other two are parity configs)")
    ap.add_argument("--crop-size", type=int, default=0, help="restore in tiles of this size (the tiled branch of the test scripts)")
    ap.add_argument("--overlap-size", type=int, default=0, help="overlap of those tiles")
    ap.add_argument("--host-io", action="store_true",
                    help="hand host buffers to the library (PCIe copies inside the timed region); not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-compare", action="store_true",
                    help="skip the second network that repeats the steps on the fp32 matrix instruction (profiling runs)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the `configs` array (the other shapes of the BASELINE metric, N = 1 only)")
    ap.add_argument("--detail", default=None, help="write the per-shape kernel table to this file")
    ap.add_argument("--images", type=int, default=0,
                    help="restore a fixed folder of this many --height x --width images (default 321 x 481), sharded by image "
                         "over the ranks and gathered on rank 0 (BASELINE configs[2]); --batch of them in flight per call")
    ap.add_argument("--train-step", action="store_true",
                    help="time the AST data-parallel optimisation step instead of the inference forward")
    ap.add_argument("--train-batch", type=int, default=2, help="crops per GPU per optimisation step (--train-step)")
    ap.add_argument("--no-graph", action="store_true",
                    help="--train-step: queue every kernel of a step from the host instead of replaying the captured step")
    ap.add_argument("--drop-path", type=float, default=None,
                    help="stochastic depth rate of --train-step (default: the reference constructor's 0.1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run the process group (RCCL) is set up even for one rank, so that the
    # launcher path can be exercised on a one-GPU box
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.train_step:
        out = train_step_bench(args, rank, local_rank, world, use_dist)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    if args.images:
        out = images_bench(args, rank, local_rank, world, use_dist)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    from jirf_hip.synthetic import default_config, make_state, synthetic_images
    from basicsr.models.archs import define_network

    cfg = default_config(args.arch)
    sd = make_state(args.arch, cfg, seed=1)
    net = define_network(dict(cfg, type=args.arch))
    net.load_state_dict(sd)
    net.cuda(local_rank).eval()
    net.batch_mode = "per_image"

    B, S = args.batch, args.size
    Hh, Ww = args.height or S, args.width or S
    x_host = synthetic_images(B, Hh, Ww, seed=100 + rank, sigma=50)
    x = x_host if args.host_io else torch.from_numpy(x_host).cuda()

    # images the network does not take as they are go through the loop body of the test scripts
    # (reflect pad, forward, crop + clamp; Deraining/test_rain100L.py:147-190), all on the device.  AST only
    # takes squares with side % 128 == 0: other images are padded to the enclosing one (a choice of this
    # package; MP/s counts the unpadded pixels).
    need = {"HINT": 8, "ASTv2": 8, "FPro": 32, "AST": 128}[args.arch]
    if args.host_io and (Hh % need or Ww % need or args.crop_size or (args.arch == "AST" and Hh != Ww)):
        raise SystemExit("--host-io takes sizes the network accepts as they are")
    script_path = Hh % need != 0 or Ww % need != 0 or (args.arch == "AST" and Hh != Ww) or args.crop_size > 0
    if script_path:
        from jirf_hip import harness

    def step():
        if args.arch == "HINT":
            net.qv_cache = None      # every image is a cold-start call
        if args.crop_size:      # tiles are independent units: pad to 8 like the scripts, split, restore, mean-blend
            return harness.restore_image(net, x, factor=8, crop_size=args.crop_size, overlap_size=args.overlap_size)
        if script_path:
            return harness.restore_image(net, x, factor=need, square_multiple=128 if args.arch == "AST" else None)
        return net(x)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(np.isfinite(y).all() if args.host_io else torch.isfinite(y).all()), "non-finite output"

    mp_per_step = world * B * Hh * Ww / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    value = mp_per_step / (ms_per_step / 1e3)
    flops_img = hint_flops_per_image(cfg, Hh, Ww) if args.arch == "HINT" else 0.0
    out = {
        "metric": "restored megapixels/sec (node), HINT sigma=50 denoising forward, 256x256 crops"
                  if (args.arch, Hh, Ww) == ("HINT", 256, 256) else
                  f"restored megapixels/sec (node), {args.arch} forward, {Ww}x{Hh} images",
        "value": round(value, 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "ms_per_image": round(ms_per_step / B, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (uniform [0,1) RGB + N(0,(50/255)^2) noise; seeded random weights in the reference key layout)"
                + ("; inputs and outputs in HOST memory, PCIe copies timed" if args.host_io else ""),
        "config": {"workload": (f"HINT (dim 48, blocks [4,6,6,8], refine 4, ffn 2.66) forward on a batch of {B} "
                                f"{Ww}x{Hh} RGB crops per GPU, cold-start per image, fp32") if args.arch == "HINT" else
                               f"{args.arch} (reference default constructor) forward on a batch of {B} {Ww}x{Hh} RGB images per GPU, fp32"
                               + ((f", reflect-padded to the enclosing {harness.square_side(Hh, Ww)}-pixel square, cropped and clamped after"
                                   if args.arch == "AST" else f", reflect-padded to multiples of {need}, cropped and clamped after")
                                  if script_path and not args.crop_size else
                                  (f", reflect-padded to multiples of 8, restored as {args.crop_size}x{args.crop_size} tiles with "
                                   f"overlap {args.overlap_size} and mean-blended" if args.crop_size else "")),
                   "batch_per_gpu": B, "height": Hh, "width": Ww, "sharding": "images round-robin over ranks, no collective"},
    }
    import jirf_hip as _jh
    split_on = bool(_jh.lib().jirf_set_gemm_split_bf16(1))
    _jh.lib().jirf_set_gemm_split_bf16(int(split_on))
    out["config"]["gemm_arithmetic"] = (
        "fp32 in, fp32 out, fp32 accumulation; matrix-bound GEMMs with shared weights form each fp32 product from three "
        "exact bf16 terms per operand (six of the nine partial products, each exact) on v_mfma_f32_32x32x16_bf16 - error "
        "against float64 at or below the fp32 matrix instruction's (tests/test_split_bf16_gpu.py); the other contractions "
        "use v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64" if split_on else
        "v_mfma_f32_32x32x2_f32 (fp32 matrix instruction) for every fp32 contraction; v_mfma_f64_16x16x4_f64 for the Gram matrix")
    if rank == 0 and world == 1 and split_on and not args.host_io and not script_path and not args.no_fp32_compare:
        # the same steps with every GEMM on the fp32 matrix instruction (a second network, packed under that setting)
        _jh.lib().jirf_set_gemm_split_bf16(0)
        try:
            net32 = define_network(dict(cfg, type=args.arch))
            net32.load_state_dict(sd)
            net32.cuda(local_rank).eval()
            net32.batch_mode = "per_image"

            def step32():
                if args.arch == "HINT":
                    net32.qv_cache = None
                return net32(x)
            for _ in range(args.warmup):
                step32()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                y32 = step32()
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - t0) / args.steps * 1e3
            rec32 = {"value": round(mp_per_step / (ms32 / 1e3), 3), "unit": "MP/s", "ms_per_step": round(ms32, 3),
                     "ms_per_image": round(ms32 / B, 4), "steps": args.steps, "warmup": args.warmup,
                     "what": "the same batch and step count with JIRF_GEMM_SPLIT_BF16=0: every GEMM on v_mfma_f32_32x32x2_f32"}
            if args.arch == "HINT":
                # HINT ranks channels by an argsort of correlations (hint_arch.py:241-296): rounding-level differences
                # reorder near-ties, and the output is discontinuous in the order.  So the two arithmetics are compared
                # once as they run and once with the fp32 network on the channel orders the headline network chose.
                p_split, p_32 = net.last_perms(groups=B), net32.last_perms(groups=B)
                rec32["free_running"] = {
                    "max_abs_diff_of_outputs": float((y32 - y).abs().max()),
                    "images_x_attention_blocks_with_a_different_channel_order":
                        int(sum(int((a_ != b_).any(axis=1).sum()) for a_, b_ in zip(p_split, p_32))),
                    "of": int(sum(a_.shape[0] for a_ in p_split))}
                net32.force_perms(p_split)
                rec32["same_channel_orders"] = {"max_abs_diff_of_outputs": float((step32() - y).abs().max())}
                net32.force_perms(None)
            else:
                rec32["max_abs_diff_of_outputs"] = float((y32 - y).abs().max())
            out["fp32_mfma_gemm"] = rec32
            del net32, y32
        finally:
            _jh.lib().jirf_set_gemm_split_bf16(1)
    if flops_img and not script_path:
        out["model_tflops_per_s"] = round(world * B * flops_img / (ms_per_step / 1e3) / 1e12, 2)
        out["frac_of_f32_mfma_peak"] = round(B * flops_img / (ms_per_step / 1e3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)

    if rank == 0 and world == 1 and not args.no_roofline and not args.host_io:
        nat = net.native()
        nat.enable_kernel_timing(True)
        acc, detail = {}, {}
        for _ in range(args.steps):
            step()
            torch.cuda.synchronize()
            for k in nat.kernel_times():
                # labels are "<kernel family> <shape>": the roofline leg aggregates per family
                for table, key in ((acc, k["name"].split(" ")[0]), (detail, k["name"])):
                    a = table.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                    for f in ("ms", "flops", "bytes", "launches"):
                        a[f] += k[f]
        nat.enable_kernel_timing(False)
        if args.detail:
            with open(args.detail, "w") as fh:
                for k, v in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
                    fh.write(f"{v['ms'] / args.steps:9.3f} ms/step  {v['launches'] // args.steps:4d} launches  "
                             f"{v['flops'] / (v['ms'] / 1e3) / 1e12 if v['ms'] else 0:7.2f} TFLOP/s  "
                             f"{v['bytes'] / (v['ms'] / 1e3) / 1e9 if v['ms'] else 0:8.1f} GB/s  {k}\n")
        tot_ms = sum(a["ms"] for a in acc.values())
        # the dominant kernel = the single launch shape with the largest share of device time
        name, roof = _dominant(detail)
        a = detail[name]
        per_launch_s = a["ms"] / 1e3 / a["launches"]
        traffic, measured_on = pmc_traffic(name, B)
        roof.update({"traffic": traffic, "traffic_measured_on": measured_on,
                     "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                     "algorithmic_flops_per_launch": a["flops"] / a["launches"],
                     "launches_per_step": a["launches"] // args.steps,
                     "avg_launch_us": round(per_launch_s * 1e6, 2),
                     "share_of_device_time": round(a["ms"] / tot_ms, 4)})
        out["roofline"] = roof
        out["kernels"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                              "launches_per_step": v["launches"] // args.steps,
                              "tflops": round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) if v["ms"] > 0 else 0,
                              "gbs": round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) if v["ms"] > 0 else 0}
                          for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"])}

    # the other configurations BASELINE.json's metric names, each timed on this GPU (N = 1, default run only)
    if rank == 0 and world == 1 and not args.no_extra_configs and not args.host_io and \
            (args.arch, Hh, Ww, args.crop_size) == ("HINT", 256, 256, 0):
        nets = {"HINT": net}
        # (arch, batch, height, width, steps, tile, overlap, carried cache)
        extra = [("HINT", 16, 256, 256, 6, 0, 0, False), ("HINT", 1, 256, 256, 20, 0, 0, False),
                 ("HINT", 1, 720, 1280, 4, 0, 0, False), ("HINT", 1, 720, 1280, 2, 256, 200, True),
                 ("AST", 16, 256, 256, 6, 0, 0, False), ("AST", 1, 256, 256, 20, 0, 0, False),
                 ("AST", 1, 321, 481, 10, 0, 0, False), ("ASTv2", 1, 321, 481, 10, 0, 0, False),
                 ("FPro", 1, 460, 620, 6, 256, 158, False)]
        out["configs"] = []
        for a_, b_, h_, w_, k_, c_, o_, cc_ in extra:
            out["configs"].append(run_extra_config(nets, a_, b_, h_, w_, k_, warmup=1 if cc_ else 2, crop=c_, overlap=o_,
                                                   carry=cc_))
            if a_ != "HINT" and a_ in nets and all(e[0] != a_ for e in extra[extra.index((a_, b_, h_, w_, k_, c_, o_, cc_)) + 1:]):
                del nets[a_]            # free the arena before the next network plans its own
        net.batch_mode = "per_image"
        # the second half of BASELINE configs[4]: one AST optimisation step (forward, loss, backward, clip, Adam)
        targs = argparse.Namespace(**dict(vars(args), train_batch=2, size=256, steps=4, warmup=2, no_cpu_baseline=True,
                                          no_roofline=False, detail=None))
        trec = train_step_bench(targs, 0, local_rank, 1, False)
        out["configs"].append({"arch": "AST", "workload": "optimisation step: " + trec["config"]["workload"],
                               "batch": 2, "height": 256, "width": 256, "steps": 4, "ms_per_step": trec["ms_per_step"],
                               "ms_per_image": round(trec["ms_per_step"] / 2, 3), "mp_per_s": trec["value"],
                               "device_ms_per_step": trec.get("device_ms_per_step"),
                               "dominant_kernel": trec.get("roofline")})

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: one 256x256 image through the oracle on the host cores of this box
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        torch.set_num_threads(cores)
        xh = torch.from_numpy(x_host)
        # the only place the benchmark touches oracle/: the CPU restatement, as the reported baseline
        if args.arch == "HINT":
            from oracle.hint_oracle import hint_forward as _f
            oracle_forward = lambda sd_, x_, cfg_: _f(sd_, x_, cfg_, None)
        elif args.arch == "ASTv2":
            from oracle.astv2_oracle import astv2_forward as oracle_forward
        elif args.arch == "FPro":
            from oracle.fpro_oracle import fpro_forward as oracle_forward
        else:
            from oracle.ast_oracle import ast_forward as oracle_forward
        # bounded sample: images of the benchmark batch, one B = 1 call each (what the step does per
        # image), until about 12 s of CPU work or the batch is used up
        t0 = time.perf_counter()
        n_img = 0
        while n_img < B and (n_img == 0 or time.perf_counter() - t0 < 12.0):
            oracle_forward(sd, xh[n_img:n_img + 1], cfg)
            n_img += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n_img * Hh * Ww / 1e6 / dt, 5), "unit": "MP/s", "cores": cores, "kind": "port",
                               "sample": f"{n_img} image(s) {Ww}x{Hh} of the benchmark batch through the oracle/ restatement "
                                         f"of {args.arch} (torch-CPU fp32, {cores} threads), {dt:.1f} s"}

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
