#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 300 python tools/attn_bench.py 64 16 512 2>&1 | grep -v "^W2026\|^E2026"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "attention or keep_masks or t5_large_L512 or reproducible" 2>&1 | grep -v "^W2026\|^E2026" | tail -4
timeout 600 python tools/leg_ab.py c5 base=norm_bwd_fuse:1 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call12.txt
