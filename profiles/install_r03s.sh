#!/bin/bash
# Copies the summaries the two collection scripts left under gpurun_out/ into profiles/ under the r03s_ names.
set -e
S=gpurun_out/prof_r03s; P=gpurun_out/prof_r03
cp $S/summary/r03s_hint_b32_kernel_stats.csv $S/summary/r03s_hint_b32_kernel_stats_by_grid.csv $S/summary/r03s_pmc_traffic.json profiles/
cp $S/summary/traffic_by_kernel.md profiles/r03s_hint_b32_traffic_by_kernel.md
cp $S/bench_under_rocprof.json profiles/r03s_hint_b32_bench_under_rocprof.json
for a in AST ASTv2 FPro; do
  l=$(echo $a | tr A-Z a-z)
  cp $P/${a}_detail.txt profiles/r03s_${l}_per_shape_hip_events.txt
  cp $P/${a}_kernel_by_grid.txt profiles/r03s_${l}_kernel_stats_by_grid.txt
  cp $P/${a}_traffic_by_kernel.md profiles/r03s_${l}_traffic_by_kernel.md
  cp $P/${a}_bench.json profiles/r03s_${l}_bench.json
  cp $P/${a}_bench_under_rocprof.json profiles/r03s_${l}_bench_under_rocprof.json
done
cp $P/HINT_b1_detail.txt profiles/r03s_hint_b1_per_shape_hip_events.txt
cp $P/AST_b1_detail.txt profiles/r03s_ast_b1_per_shape_hip_events.txt
cp $P/HINT_b1_kernel_by_grid.txt profiles/r03s_hint_b1_kernel_stats_by_grid.txt
cp $P/HINT_b32_sq_by_kernel.txt profiles/r03s_hint_b32_sq_counters_by_kernel.txt
cp $P/train_kernel_by_grid.txt profiles/r03s_ast_train_step_kernel_stats_by_grid.txt
if [ -f gpurun_out/r03s_bench_default.json ]; then
  cp gpurun_out/r03s_bench_default.json profiles/r03s_hint_b32_bench_default.json
  cp gpurun_out/r03s_detail.txt profiles/r03s_hint_b32_per_shape_hip_events.txt
  python3 profiles/make_roofline_table.py profiles/r03s_hint_b32_per_shape_hip_events.txt > profiles/r03s_roofline_table.md
fi
