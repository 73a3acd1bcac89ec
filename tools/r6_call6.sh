#!/bin/bash
# Round-6 call 6: T5LayerNorm backward in the data-gradient GEMM epilogues -- parity on the hardware (op level, C2 gradients, reproducibility), in-step A/B.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k "norm_backward or row_sums or wave_specialised or bf16_gradients_at_benchmark or test_model_bf16 or reproducible or gated or resume" 2>&1 | grep -v "^W2026\|^E2026" | tail -25
timeout 600 python tools/train_ab6.py --show gemm5,gemm2,p5_gemm_kernel,rmsnorm_bwd,attn_bwd_fused,reduce_rows old=norm_bwd_fuse:0,gemm_ws128:0 ws128=gemm_ws128:1 fuse=norm_bwd_fuse:1,gemm_ws128:1 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call6.txt
