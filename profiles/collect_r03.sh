#!/bin/bash
# Profiles of the three window-attention networks, the AST optimisation step and the B = 1 call.
# Run on the GPU box from the repository root:  bash profiles/collect_r03.sh
# Per network: a kernel trace, the per-shape HIP-event table of an unprofiled run, and the two traffic passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Every GPU step has its own time limit; the first failure ends the run.
set -e
out=gpurun_out/prof_r03
mkdir -p $out
export TMPDIR=/tmp
common="--no-cpu-baseline --no-extra-configs --no-fp32-compare"
run_arch() {
  a=$1; shift
  timeout -k 10 200 python3 bench.py --arch $a "$@" --steps 5 --warmup 2 $common --detail $out/${a}_detail.txt > $out/${a}_bench.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${a}_trace -o bench -- python3 bench.py --arch $a "$@" --steps 3 --warmup 1 $common > $out/${a}_bench_under_rocprof.json
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_${a}_fetch -o pmc -- python3 bench.py --arch $a "$@" --steps 1 --warmup 1 $common --no-roofline > $out/${a}_under_pmc_fetch.json
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_${a}_write -o pmc -- python3 bench.py --arch $a "$@" --steps 1 --warmup 1 $common --no-roofline > $out/${a}_under_pmc_write.json
  echo "$a done"
}
run_arch AST --batch 16
run_arch ASTv2 --height 328 --width 488 --batch 8
run_arch FPro --batch 16
# the optimisation step (window attention backward)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/train_trace -o train -- python3 bench.py --train-step --steps 3 --warmup 1 --no-cpu-baseline > $out/train_under_rocprof.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_train_fetch -o pmc -- python3 bench.py --train-step --no-graph --steps 1 --warmup 1 --no-cpu-baseline > $out/train_under_pmc_fetch.json
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_train_write -o pmc -- python3 bench.py --train-step --no-graph --steps 1 --warmup 1 --no-cpu-baseline > $out/train_under_pmc_write.json
echo "train done"
# the single-image call of the two benchmark networks
timeout -k 10 200 python3 bench.py --arch HINT --batch 1 --steps 20 --warmup 3 $common --detail $out/HINT_b1_detail.txt > $out/HINT_b1_bench.json
timeout -k 10 200 python3 bench.py --arch AST --batch 1 --steps 20 --warmup 3 $common --detail $out/AST_b1_detail.txt > $out/AST_b1_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/HINT_b1_trace -o bench -- python3 bench.py --arch HINT --batch 1 --steps 5 --warmup 1 $common > $out/HINT_b1_under_rocprof.json
# SQ counters of every kernel of the HINT step (B = 32)
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $out/sq/pmc_HINT_sq -o pmc -- python3 bench.py --steps 1 --warmup 1 $common --no-roofline > $out/HINT_under_pmc_sq.json
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $out/sq/pmc_HINT_sq2 -o pmc -- python3 bench.py --steps 1 --warmup 1 $common --no-roofline > $out/HINT_under_pmc_sq2.json
# summaries (the databases themselves are too large to bring back)
for a in AST ASTv2 FPro; do
  python3 profiles/train_by_grid.py $(ls $out/${a}_trace/*.db | head -1) 3 80 > $out/${a}_kernel_by_grid.txt
  python3 profiles/traffic_by_kernel.py $out pmc_${a} > $out/${a}_traffic_by_kernel.md
done
python3 profiles/train_by_grid.py $(ls $out/train_trace/*.db | head -1) 3 80 > $out/train_kernel_by_grid.txt
python3 profiles/traffic_by_kernel.py $out pmc_train > $out/train_traffic_by_kernel.md
python3 profiles/train_by_grid.py $(ls $out/HINT_b1_trace/*.db | head -1) 5 120 > $out/HINT_b1_kernel_by_grid.txt
python3 profiles/exp/pmc_sum.py $out/sq > $out/HINT_b32_sq_by_kernel.txt
find $out -name "*.db" -delete
echo "all done"
