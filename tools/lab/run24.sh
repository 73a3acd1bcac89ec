#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t24_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t24_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t24_bench.txt 2>&1
}
run bench.py P5_WW_GATHER=1
run bench.py P5_WW_GATHER=0
run bench.py P5_WW_GATHER=1
run bench.py P5_WW_GATHER=0
cat gpurun_out/t24_bench.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "test_model or reproducible or stored_not" 2>&1 | tail -2
