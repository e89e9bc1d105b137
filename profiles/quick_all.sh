#!/bin/bash
# quick check used while tuning: full GPU suite, then the four architectures' benchmarks (one line each)
cd /root/repo
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for a in HINT ASTv2 AST FPro; do
  extra=""; if [ $a = ASTv2 ]; then extra="--height 328 --width 488 --batch 8"; fi
  timeout -k 10 200 python bench.py --arch $a $extra --steps 5 --warmup 2 --no-cpu-baseline --detail gpurun_out/detail_$a.txt > gpurun_out/bench_$a.json || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$a.json"))
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])
print("$a", d["value"], "MP/s", d["ms_per_step"], "ms/step", {k:v["ms_per_step"] for k,v in ks if "dwconv" in k or "win" in k})
PY
done
