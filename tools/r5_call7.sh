#!/bin/bash
# round 5, call 7: full GPU suite (incl. both dataset gates), default bench line, split GEMM probe rerun
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call7.txt; : > $OUT
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -9 | tee -a $OUT
timeout 2400 python -m pytest tests -q -m gpu -s --durations=15 > gpurun_out/r5_gpu_suite_full.log 2>&1
grep "\[dataset\]\|\[verified\|passed\|failed\|FAILED\|slowest\|^[0-9.]*s call" gpurun_out/r5_gpu_suite_full.log | cut -c1-700 | tee -a $OUT
timeout 900 python bench.py > gpurun_out/r5_bench_full.log 2>&1
grep '^{' gpurun_out/r5_bench_full.log | tail -1 > gpurun_out/r5_bench_line.json
python - <<'PY' | tee -a $OUT
import json
l = json.load(open("gpurun_out/r5_bench_line.json"))
print("ms/step", l["ms_per_step"], "samples/s", l["value"], "roofline", {k: l["roofline"].get(k) for k in ("kernel", "frac", "frac_excl_dispatch", "bracket_floor_us", "us_per_step")})
print("generation", {k: l["generation"].get(k) for k in ("items_per_s", "ms_per_batch", "ms_per_batch_median_call", "verify_stats")})
print("plain bf16", {k: l["generation_plain_bf16"].get(k) for k in ("items_per_s", "ms_per_batch", "timing_ms")})
print("roofline_generation", l.get("roofline_generation"))
for k, v in (l.get("legs") or {}).items(): print("leg", k, v)
print("cpu", l.get("cpu_baseline"), l.get("cpu_baseline_generation"))
PY
