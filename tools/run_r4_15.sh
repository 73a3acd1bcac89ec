#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONDONTWRITEBYTECODE=1
timeout 75 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/final_smoke.log 2>&1; tail -3 gpurun_out/final_smoke.log
