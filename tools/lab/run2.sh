#!/bin/bash
# GPU call 2 of round 3: lab2 (128x128 wave tiles, two WGs/CU, K rotation, grouped wgrad variants), the bf16 T5-base gradient
# diagnostic, the new kernel tests, the engine with grouped weight gradients (parity + step-time A/B).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 tools/lab/gemm_lab lab2 > gpurun_out/lab2.txt 2>&1; echo "lab rc $?" >> gpurun_out/lab2.txt
tail -3 gpurun_out/lab2.txt
timeout 600 python tools/diag_r3.py t5-base 2 8 128 8 > gpurun_out/diag_base2.txt 2>&1; tail -4 gpurun_out/diag_base2.txt
timeout 900 python tools/diag_r3.py t5-base 12 8 128 8 > gpurun_out/diag_base12.txt 2>&1; tail -4 gpurun_out/diag_base12.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "persistent_ring or test_model or golden or trajectory or fused_loss or bf16_gradients_at_benchmark or dropout_on or test_gemm" > gpurun_out/t2_parity.log 2>&1; echo "rc $?" >> gpurun_out/t2_parity.log
tail -3 gpurun_out/t2_parity.log
for v in "P5_WGRAD_GROUP=0" "P5_WGRAD_GROUP=1 P5_G4_NST=3" "P5_WGRAD_GROUP=1 P5_G4_NST=5" "P5_WGRAD_GROUP=1 P5_G4_NST=2" "P5_WGRAD_GROUP=0" "P5_WGRAD_GROUP=1 P5_G4_NST=3"; do
  echo "== $v" >> gpurun_out/t2_bench.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t2_bench.txt 2>&1
done
cat gpurun_out/t2_bench.txt
