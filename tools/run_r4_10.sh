#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 500 python tools/gate_explore.py chain 30 3e-4 3 > gpurun_out/r4/gate_chain30_lr3e-4.txt 2>&1; grep -v amdgpu gpurun_out/r4/gate_chain30_lr3e-4.txt | tail -7 | cut -c1-330
timeout 500 python tools/gate_explore.py chain 60 2e-4 3 > gpurun_out/r4/gate_chain60_lr2e-4.txt 2>&1; grep -v amdgpu gpurun_out/r4/gate_chain60_lr2e-4.txt | tail -7 | cut -c1-330
