#!/bin/bash
# Round-6 call 5: wave-specialised 128x128 instance for the N = d_model GEMMs (parity on the hardware, in-step A/B incl. K = 512), workgroup count of the norm backward.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wave_specialised or bf16_gradients_at_benchmark or test_model_bf16 or reproducible" 2>&1 | grep -v "^W2026\|^E2026" | tail -6
timeout 600 python tools/train_ab6.py old=gemm_ws128:0,norm_bwd_blocks:1024,gemm_ws128_min_k:512 ws128=gemm_ws128:1 ws128k1024=gemm_ws128:1,gemm_ws128_min_k:1024 nb512=norm_bwd_blocks:512 nb256=norm_bwd_blocks:256 both=gemm_ws128:1,norm_bwd_blocks:512 2>&1 | grep "ms/step"
timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
P5_NORM_BWD_BLOCKS=512 timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
P5_NORM_BWD_BLOCKS=256 timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
} 2>&1 | tee gpurun_out/r6_call5.txt
