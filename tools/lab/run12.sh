#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t12_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t12_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t12_bench.txt 2>&1
}
run bench.py P5_AUX_STREAM=1
run bench.py P5_AUX_STREAM=0
run bench.py P5_AUX_STREAM=1
run bench.py P5_AUX_STREAM=0
cat gpurun_out/t12_bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_ddp.py -q -x -k "trajectory or training_converges or runner or resume or two_ranks or world2" > gpurun_out/t12_tests.log 2>&1; tail -2 gpurun_out/t12_tests.log
