#!/bin/bash
# Collects the profiles kept in this directory.  Run on the GPU box from the repository root:
#   bash profiles/collect.sh <tag>        (tag e.g. r01)
# 1. rocprofv3 --kernel-trace --stats over the default benchmark command (short step count)
# 2. over the dominant launch shape alone (profiles/pmc_op.py): two PMC passes (reads, writes:
#    they do not fit one pass), one SQ pass (MFMA busy and wait cycles) and one kernel-trace pass
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# the isolated launches run the form the networks run (LayerNorm affine folded into the packed weights)
export JIRF_OP_LN_FOLD=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-fp32-compare > $out/bench_under_rocprof.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 profiles/pmc_op.py
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 profiles/pmc_op.py
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc_sq -o pmc -- python3 profiles/pmc_op.py
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/op_trace -o op -- python3 profiles/pmc_op.py
# 3. the whole benchmark step under the two traffic counters (one step: every launch shape of the network once
#    per image), for the per-kernel / per-grid traffic table (profiles/traffic_by_kernel.py)
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_bench_fetch -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs --no-fp32-compare > $out/bench_under_pmc_fetch.json
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_bench_write -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs --no-fp32-compare > $out/bench_under_pmc_write.json
# 4. the AST optimisation step
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/train_trace -o train -- python3 bench.py --train-step --steps 3 --warmup 1 --no-cpu-baseline > $out/train_under_rocprof.json
find $out -type f | head -80
