#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t7_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t7_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_fwd']['frac'])" >> gpurun_out/t7_bench.txt 2>&1
}
run bench.py A=1
run bench.py P5_WGRAD_LAYERS=2
run bench.py P5_WGRAD_LAYERS=2 P5_G4_NST=2
run bench.py P5_G4_NST=2
run bench.py P5_GEMM_WIDE=0
run bench.py P5_GEMM_RING_N512=0
run tools/bench_side.py A=1
run bench.py A=1
cat gpurun_out/t7_bench.txt
timeout 900 python bench.py --steps 20 --warmup 5 --legs none > gpurun_out/t7_full.log 2>&1; grep '^{' gpurun_out/t7_full.log | tail -1 > gpurun_out/t7_full.json; python -c "
import json; d=json.load(open('gpurun_out/t7_full.json')); print({k:d[k] for k in ('ms_per_step','value','beam10_items_per_sec')}); print(d['generation']); print(d['roofline_generation']); print(d['cpu_baseline'])"
