#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/lab/gemm_lab lab4 > gpurun_out/lab4c.txt 2>&1
grep -v "M=8000\|mode" gpurun_out/lab4c.txt | grep -v "^  g4\|g5 abl" | head -70
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "loader_drained" > gpurun_out/t19_tests.log 2>&1; tail -5 gpurun_out/t19_tests.log
rm -f gpurun_out/t19_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t19_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t19_bench.txt 2>&1
}
run bench.py P5_GEMM_WS=3
run bench.py P5_GEMM_WS=7
run bench.py P5_GEMM_WS=3
run bench.py P5_GEMM_WS=7
cat gpurun_out/t19_bench.txt
