#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/lab/gemm_lab lab4 > gpurun_out/lab4.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wave_specialised" > gpurun_out/t14_tests.log 2>&1; tail -3 gpurun_out/t14_tests.log
rm -f gpurun_out/t14_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t14_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t14_bench.txt 2>&1
}
run bench.py P5_GEMM_WS=0
run bench.py P5_GEMM_WS=1
run bench.py P5_GEMM_WS=0
run bench.py P5_GEMM_WS=1
cat gpurun_out/t14_bench.txt
cat gpurun_out/lab4.txt
