#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t8_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t8_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'], d['generation']['ms_per_batch'], d['generation']['timing_ms'])" >> gpurun_out/t8_bench.txt 2>&1
}
run bench.py P5_NORM_FUSE=1
run bench.py P5_NORM_FUSE=0
run bench.py P5_NORM_FUSE=1
run bench.py P5_NORM_FUSE=0
cat gpurun_out/t8_bench.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "test_model or golden or trajectory or fused_loss or bf16_gradients or generate or bf16_training" > gpurun_out/t8_parity.log 2>&1; tail -3 gpurun_out/t8_parity.log
timeout 600 python -m pytest tests/test_gpu_dataset.py -q -s 2>&1 | grep -E "^\[dataset\] bf16|passed|failed|^E " | grep -v "tie report" | cut -c1-300
