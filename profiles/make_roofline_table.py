"""Per-launch-shape roofline table from the HIP-event table bench.py writes with --detail.
Usage: python profiles/make_roofline_table.py profiles/r01_hint_b16_per_shape_hip_events.txt > profiles/r01_roofline_table.md
Every launch shape is priced against both roofs (fp32 matrix peak 157.3 TFLOP/s, HBM 8 TB/s) with
its algorithmic FLOPs and bytes; the larger fraction names the roof that bounds it."""
import sys

MFMA, HBM = 157.3, 8000.0
rows = []
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) < 9:
        continue
    ms, launches, tf, gbs, name = float(f[0]), int(f[2]), float(f[4]), float(f[6]), " ".join(f[8:])
    fm, fh = tf / MFMA, gbs / HBM
    matrix = name.startswith("gemm") or "chan_stats" in name
    bound = "mfma" if (matrix and fm >= fh) else "hbm"
    rows.append((ms, launches, name, tf, gbs, bound, fm if bound == "mfma" else fh))
tot = sum(r[0] for r in rows)
print("| ms/step | share | launches | kernel + shape | TFLOP/s | GB/s | bound | fraction of roof |")
print("|---:|---:|---:|---|---:|---:|---|---:|")
for ms, n, name, tf, gbs, bound, frac in rows:
    print(f"| {ms:.3f} | {100 * ms / tot:.1f} % | {n} | `{name}` | {tf:.1f} | {gbs:.0f} | {bound} | {frac:.2f} |")
print(f"\nTotal {tot:.1f} ms of kernel time per step.")
