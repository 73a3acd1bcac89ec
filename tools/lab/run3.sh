#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/diag_r3.py t5-base 12 8 128 8 > gpurun_out/diag3_base12.txt 2>&1; grep -c "tensors with relL2" gpurun_out/diag3_base12.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "persistent_ring or test_model or golden or trajectory or fused_loss or bf16_gradients or test_gemm or attention" > gpurun_out/t3_parity.log 2>&1; echo "rc $?" >> gpurun_out/t3_parity.log
tail -3 gpurun_out/t3_parity.log
rm -f gpurun_out/t3_bench.txt
for v in "P5_GEMM_WIDE=0 P5_GEMM_RING_N512=0" "P5_GEMM_WIDE=1 P5_GEMM_RING_N512=0" "P5_GEMM_WIDE=0 P5_GEMM_RING_N512=1" "P5_GEMM_WIDE=1 P5_GEMM_RING_N512=1" "P5_GEMM_WIDE=0 P5_GEMM_RING_N512=0" "P5_GEMM_WIDE=1 P5_GEMM_RING_N512=1" "P5_GEMM_WIDE=1 P5_GEMM_RING_N512=1 P5_GEMM_WIDE_MIN_TILES=100"; do
  echo "== $v" >> gpurun_out/t3_bench.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t3_bench.txt 2>&1
done
cat gpurun_out/t3_bench.txt
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu --no-gen --legs configs 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:(v.get('ms_per_step'), v.get('ms_per_batch')) for k,v in d['legs'].items()}))" > gpurun_out/t3_legs.txt 2>&1; cat gpurun_out/t3_legs.txt
