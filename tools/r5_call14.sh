#!/bin/bash
# Round-5 call 14: the head-resident attention kernels (128 < L <= 512) on the hardware: parity tests, isolated timing at the C5 shape,
# the C5 step with them on / off.   gpurun --timeout 900 -- 'bash tools/r5_call14.sh'
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "attention or L512 or bench_generation_timing or t5_large" 2>&1 | tail -5
timeout 200 python tools/attn_bench.py 64 16 512 2>&1 | grep -v "^W2026\|^E2026" | tail -8
for HEAD in "1 1" "1 0" "0 0"; do
  set -- $HEAD
  echo "C5 step, P5_ATTN_FWD_HEAD=P5_ATTN_BWD_HEAD=$1 P5_ATTN_KEEP_BITS=$2"
  P5_ATTN_FWD_HEAD=$1 P5_ATTN_BWD_HEAD=$1 P5_ATTN_KEEP_BITS=$2 timeout 300 python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu 2>&1 | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.read()); print('ms/step', l['ms_per_step'], 'samples/s', l['value'])
for c in l.get('step_kernels', [])[:8]: print('   ', c['kernel'][:70], c['launches_per_step'], round(c['us_per_step']), c.get('tflops'))
"
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "reproducible" 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r5_call14.txt
