#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call11.txt; : > $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_ddp.py -x -q -k "lanes or runner or filtered or world2 or two_ranks or resume" 2>&1 | tail -5 | tee -a $OUT
timeout 600 python bench.py --no-cpu --legs none --steps 5 --warmup 2 > gpurun_out/r5_bench_lanes.log 2>&1
grep '^{' gpurun_out/r5_bench_lanes.log | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.read())
g = l['generation']; d = l['generation_plain_bf16']
print('verified', {k: g.get(k) for k in ('items_per_s', 'ms_per_batch', 'ms_per_batch_median_call', 'lanes', 'verify_stats')})
print('after noise', g.get('after_noise_training'))
print('plain', {k: d.get(k) for k in ('items_per_s', 'ms_per_batch', 'ms_per_batch_median_call')})
" | tee -a $OUT
tail -3 gpurun_out/r5_bench_lanes.log | cut -c1-300
