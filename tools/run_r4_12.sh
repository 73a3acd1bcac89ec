#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/repro
export PYTHONDONTWRITEBYTECODE=1
for p in a b; do timeout 400 python tools/diag_repro.py pipe p_$p 2 2>&1 | grep -v amdgpu.ids | tail -24; done > gpurun_out/repro/pipe.txt
python tools/diag_repro.py cmp gpurun_out/repro/pipe_p_*.pt >> gpurun_out/repro/pipe.txt 2>&1
rm -f gpurun_out/repro/*.pt
cut -c1-260 gpurun_out/repro/pipe.txt
