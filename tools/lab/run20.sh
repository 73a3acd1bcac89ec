#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py -q -x -k "stored_not_accumulated or trajectory or training_converges or resume or runner" > gpurun_out/t20_tests.log 2>&1; tail -3 gpurun_out/t20_tests.log
rm -f gpurun_out/t20_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t20_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t20_bench.txt 2>&1
}
run bench.py P5_GRAD_STORE_FIRST=1
run bench.py P5_GRAD_STORE_FIRST=0
run bench.py P5_GRAD_STORE_FIRST=1
run bench.py P5_GRAD_STORE_FIRST=0
cat gpurun_out/t20_bench.txt
