#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t5_bench.txt
run() {  # label, script, env...
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t5_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t5_bench.txt 2>&1
}
run bench.py A=1
run bench.py P5_WGRAD_LAYERS=2
run bench.py P5_WGRAD_LAYERS=2 P5_G4_NST=2
run bench.py P5_WGRAD_WGS=128
run bench.py P5_WGRAD_WGS=96
run tools/bench_noside.py A=1
run tools/bench_noside.py P5_WGRAD_LAYERS=2 P5_G4_NST=2
run tools/bench_noside.py P5_WGRAD_LAYERS=2
run bench.py P5_GEMM_WIDE=0
run tools/bench_noside.py P5_GEMM_WIDE=0
run bench.py A=1
run bench.py P5_WGRAD_LAYERS=2
cat gpurun_out/t5_bench.txt
