#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONDONTWRITEBYTECODE=1
timeout 420 python -m pytest tests/test_gpu_dataset.py -q -x -m gpu -s > gpurun_out/final_pytest_dataset2.log 2>&1; grep "dataset\] bf16:\|dataset\] oracle\|teacher-forced\|passed\|failed\|^E " gpurun_out/final_pytest_dataset2.log | sed 's/.extra.: \[.*//' | cut -c1-300 | tail -14
