"""Per-launch-shape roofline table from the HIP-event table bench.py writes with --detail.
Usage: python profiles/make_roofline_table.py profiles/r01_hint_b16_per_shape_hip_events.txt > profiles/r01_roofline_table.md
Every launch shape is priced against both roofs (fp32 matrix peak 157.3 TFLOP/s, HBM 8 TB/s) with
its algorithmic FLOPs and bytes; the larger fraction names the roof that bounds it.
`hint_chan_stats` runs half of its FLOPs (the Gram matrix that feeds the channel ranking) on v_mfma_f64_16x16x4, which has
half the fp32 rate: its matrix roof is 4 / (2 / 157.3 + 2 / 78.65) = 104.9 TFLOP/s; the table prices it against that and
says so in the bound column.  Kernels labelled `_bf16x3` form each fp32 product from six bf16 partial products on
v_mfma_f32_32x32x16_bf16 (dense peak 2500 TFLOP/s): their matrix roof in algorithmic (fp32) FLOPs is 2500 / 6 = 416.7 TFLOP/s."""
import sys

MFMA, HBM = 157.3, 8000.0
rows = []
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) < 9:
        continue
    ms, launches, tf, gbs, name = float(f[0]), int(f[2]), float(f[4]), float(f[6]), " ".join(f[8:])
    mixed = name.startswith("hint_chan_stats")
    split = name.split(" ")[0].endswith("_bf16x3")
    fm, fh = tf / (104.9 if mixed else 2500.0 / 6 if split else MFMA), gbs / HBM
    matrix = name.startswith("gemm") or "chan_stats" in name
    bound = ("mfma f32+f64" if mixed else "mfma bf16 x6" if split else "mfma") if (matrix and fm >= fh) else "hbm"
    rows.append((ms, launches, name, tf, gbs, bound, fm if bound.startswith("mfma") else fh))
tot = sum(r[0] for r in rows)
print("| ms/step | share | launches | kernel + shape | TFLOP/s | GB/s | bound | fraction of roof |")
print("|---:|---:|---:|---|---:|---:|---|---:|")
for ms, n, name, tf, gbs, bound, frac in rows:
    print(f"| {ms:.3f} | {100 * ms / tot:.1f} % | {n} | `{name}` | {tf:.1f} | {gbs:.0f} | {bound} | {frac:.2f} |")
print(f"\nTotal {tot:.1f} ms of kernel time per step.")
