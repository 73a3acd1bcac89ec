#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm" > gpurun_out/t17_tests.log 2>&1; tail -3 gpurun_out/t17_tests.log
rm -f gpurun_out/t17_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t17_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t17_bench.txt 2>&1
}
run bench.py A=1
run bench.py A=1
cat gpurun_out/t17_bench.txt
bash profiles/profile.sh r03c_train python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
