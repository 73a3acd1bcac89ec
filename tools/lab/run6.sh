#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t6_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t6_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t6_bench.txt 2>&1
}
run bench.py P5_WGRAD_SIDE=1
run bench.py P5_WGRAD_SIDE=0
run bench.py P5_WGRAD_SIDE=3
run tools/bench_noside.py P5_WGRAD_SIDE=1
run bench.py P5_WGRAD_SIDE=1 P5_WGRAD_LAYERS=2
run bench.py P5_WGRAD_SIDE=1 P5_WGRAD_LAYERS=2 P5_G4_NST=2
run bench.py P5_WGRAD_SIDE=1 P5_G4_NST=2
run bench.py P5_WGRAD_SIDE=1 P5_GEMM_WIDE=0
run bench.py P5_WGRAD_SIDE=1 P5_GEMM_RING_N512=0
run bench.py P5_WGRAD_SIDE=1
cat gpurun_out/t6_bench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_model or golden or trajectory or fused_loss or bf16_gradients" > gpurun_out/t6_parity.log 2>&1; tail -2 gpurun_out/t6_parity.log
