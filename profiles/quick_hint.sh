#!/bin/bash
# quick check used while tuning: HINT parity tests, then the default benchmark with the per-shape table
set -e
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_hint_gpu.py tests/test_golden_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --detail gpurun_out/detail_HINT.txt > gpurun_out/b16.json
python - <<PY
import json
d=json.load(open("gpurun_out/b16.json"))
print(d["value"], d["ms_per_image"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
PY
grep -E "attn_tail|attn_v|M96 K96|M96 K192" gpurun_out/detail_HINT.txt || true
