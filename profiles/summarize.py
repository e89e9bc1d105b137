"""Turn the rocprofv3 (rocpd sqlite) outputs of profiles/collect.sh into the small text/JSON
summaries kept under profiles/.  Usage: python profiles/summarize.py <tag>   (reads
gpurun_out/prof_<tag>/, writes profiles/<tag>_*)."""
import json
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 32   # crops per step of the benchmark the traces were taken on
src = f"gpurun_out/prof_{tag}"
dst = os.environ.get("PROFILES_DST", "profiles")   # (on the GPU box only gpurun_out/ comes back)
os.makedirs(dst, exist_ok=True)


def short(name):
    return name if len(name) <= 150 else name[:147] + "..."


# 1. kernel-trace statistics: per kernel symbol, and per (symbol, grid) so that one launch shape
#    of a templated kernel can be read off
db = sqlite3.connect(os.path.join(src, "trace", "bench_results.db"))
rows = db.execute("select name, grid_x, grid_y, workgroup_x, duration from kernels").fetchall()
by_sym, by_shape = {}, {}
for name, gx, gy, wx, dur in rows:
    a = by_sym.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += dur
    b = by_shape.setdefault((name, gx, gy, wx), [0, 0])
    b[0] += 1
    b[1] += dur
total = sum(v[1] for v in by_sym.values())
with open(os.path.join(dst, f"{tag}_hint_b{BATCH}_kernel_stats.csv"), "w") as fh:
    fh.write("kernel,calls,total_ns,avg_ns,percent\n")
    for k, (n, t) in sorted(by_sym.items(), key=lambda kv: -kv[1][1]):
        fh.write(f"\"{short(k)}\",{n},{t},{t / n:.0f},{100.0 * t / total:.2f}\n")
with open(os.path.join(dst, f"{tag}_hint_b{BATCH}_kernel_stats_by_grid.csv"), "w") as fh:
    fh.write("kernel,grid_x,grid_y,workgroup_x,calls,total_ns,avg_ns,percent\n")
    for (k, gx, gy, wx), (n, t) in sorted(by_shape.items(), key=lambda kv: -kv[1][1]):
        fh.write(f"\"{short(k)}\",{gx},{gy},{wx},{n},{t},{t / n:.0f},{100.0 * t / total:.2f}\n")

# 1b. the same for the AST optimisation step (bench.py --train-step)
p_train = os.path.join(src, "train_trace", "train_results.db")
if os.path.exists(p_train):
    dbt = sqlite3.connect(p_train)
    agg = {}
    for name, dur in dbt.execute("select name, duration from kernels"):
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += dur
    tt = sum(v[1] for v in agg.values())
    with open(os.path.join(dst, f"{tag}_ast_train_step_kernel_stats.csv"), "w") as fh:
        fh.write("kernel,calls,total_ns,avg_ns,percent\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write(f"\"{short(k)}\",{n},{t},{t / n:.0f},{100.0 * t / tt:.2f}\n")

# 2. PMC passes over the dominant launch shape alone (profiles/pmc_op.py).
#    FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide streaming
#    reads at half their size, so reads are doubled (writes are exact).
out = {}
for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    p = os.path.join(src, which, "pmc_results.db")
    if not os.path.exists(p):
        continue
    d = sqlite3.connect(p)
    q = ("select k.name, k.grid_x, e.counter_value from pmc_events e join kernels k "
         "on e.dispatch_id = k.dispatch_id where e.counter_name = ?")
    acc = {}
    for name, gx, val in d.execute(q, (counter,)):
        a = acc.setdefault((name, gx), [0, 0.0])
        a[0] += 1
        a[1] += val
    for (name, gx), (n, v) in acc.items():
        if "gemm" not in name:
            continue
        o = out.setdefault(short(name), {"grid_x": gx, "launches": n})
        o[counter + "_KiB_per_launch"] = v / n
for o in out.values():
    rd = 2.0 * 1024.0 * o.get("FETCH_SIZE_KiB_per_launch", 0.0)
    wr = 1024.0 * o.get("WRITE_SIZE_KiB_per_launch", 0.0)
    o.update({"read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr})
iso = None
p_iso = os.path.join(src, "op_trace", "op_results.db")
if os.path.exists(p_iso):
    d = sqlite3.connect(p_iso)
    r = d.execute("select avg(duration), count(*) from kernels where name like '%gemm%'").fetchone()
    iso = {"avg_ns": r[0], "launches": r[1],
           "note": "same launch shape run alone on N(0,1) inputs (profiles/pmc_op.py); the clock the chip holds, and "
                   "with it the duration, depends on the operand data, so this differs from the in-network launches"}
sq = None
p_sq = os.path.join(src, "pmc_sq", "pmc_results.db")
if os.path.exists(p_sq):
    d = sqlite3.connect(p_sq)
    q = ("select e.counter_name, avg(e.counter_value) from pmc_events e join kernels k on e.dispatch_id = k.dispatch_id "
         "where k.name like '%gemm%' group by e.counter_name")
    c = {name: val for name, val in d.execute(q)}
    if c.get("SQ_BUSY_CYCLES"):
        # one row per shader engine: 8 CUs = 32 SIMDs each; SQ_VALU_MFMA_BUSY_CYCLES sums over them
        sq = {"counters_avg_per_shader_engine": c,
              "mfma_busy_fraction": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 32.0 / c["SQ_BUSY_CYCLES"],
              "wave_time_split": {k: c.get(k, 0.0) / c["SQ_WAVE_CYCLES"] for k in
                                  ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c.get("SQ_WAVE_CYCLES")}}
import hashlib
_h = hashlib.sha256()
for _f in ("gemm.hip", "gemm_split.hip", "split_bf16.h"):
    with open(os.path.join("image-restoration_amd", "csrc", _f), "rb") as _fh:
        _h.update(_fh.read())
with open(os.path.join("image-restoration_amd", "csrc", "common.h")) as _fh:
    _t = _fh.read()
_h.update(_t[_t.index("struct GemmArgs {"):_t.index("void launch_gemm(")].encode())
res = {"kernel_source_sha256": _h.hexdigest(), "isolated_kernel_trace": iso, "sq_counters": sq, "op": f"gemm_ln_1x1_bf16x3 M510 K96 P65536 (B={BATCH}): LayerNorm + 1x1 conv, the dominant launch shape of the HINT benchmark",
       "algorithmic_bytes_per_launch": 4.0 * BATCH * 65536 * (96 + 510),
       "kernels": out}
with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res, indent=1))
