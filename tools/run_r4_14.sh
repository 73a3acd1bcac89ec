#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONDONTWRITEBYTECODE=1
export P5_DATASET_DUMP=gpurun_out/dataset_gate_dump.pt
timeout 320 python -m pytest tests/test_gpu_dataset.py -q -x -m gpu -s > gpurun_out/final_pytest_dataset3.log 2>&1; grep "dataset\] \|passed\|failed\|^E " gpurun_out/final_pytest_dataset3.log | cut -c1-400 | tail -18
