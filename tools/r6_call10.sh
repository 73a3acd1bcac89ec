#!/bin/bash
# Round-6 call 10: norm-backward epilogue beyond the one-tile-per-CU case (option norm_bwd_fuse 2) at C3 (T5-base) and C5 (T5-large, L = 512).
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 600 python tools/leg_ab.py c3 fuse1=norm_bwd_fuse:1 fuse2=norm_bwd_fuse:2 fuse0=norm_bwd_fuse:0 2>&1 | grep "ms/step"
timeout 900 python tools/leg_ab.py c5 fuse1=norm_bwd_fuse:1 fuse2=norm_bwd_fuse:2 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call10.txt
