#!/bin/bash
# Round-6 call 7: norm-backward epilogue with n written by the loader waves -- parity (op level, C2 gradients), in-step A/B incl. side-stream settings.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "norm_backward or bf16_gradients_at_benchmark or reproducible" 2>&1 | grep -v "^W2026\|^E2026" | tail -4
timeout 900 python tools/train_ab6.py --show gemm5,rmsnorm_bwd,attn_bwd_fused old=norm_bwd_fuse:0,gemm_ws128:0,wgrad_side:1 ws128=gemm_ws128:1 fuse=norm_bwd_fuse:1,gemm_ws128:1 fuse_s3=norm_bwd_fuse:1,gemm_ws128:1,wgrad_side:3 nofuse_s3=norm_bwd_fuse:0,gemm_ws128:1,wgrad_side:3 fuse_s0=norm_bwd_fuse:1,gemm_ws128:1,wgrad_side:0 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call7.txt
