#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONDONTWRITEBYTECODE=1
timeout 66 python -m pytest tests/test_gpu_runner.py tests/test_gpu_parity.py -q -x -m gpu -k "resume_on_device or backward_is_reproducible" > gpurun_out/final_repro_rebuilt.log 2>&1; tail -3 gpurun_out/final_repro_rebuilt.log
