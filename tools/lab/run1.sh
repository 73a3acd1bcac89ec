#!/bin/bash
# GPU call 1 of round 3: GEMM lab (every variant on the step's shapes + ablations), then the new / changed parity tests.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 tools/lab/gemm_lab all > gpurun_out/lab1.txt 2>&1; echo "lab rc $?" >> gpurun_out/lab1.txt
tail -5 gpurun_out/lab1.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k "persistent_ring or test_model_bf16 or bf16_gradients or generate_bf16_ranked_set" > gpurun_out/t1_parity.log 2>&1; echo "rc $?" >> gpurun_out/t1_parity.log
tail -3 gpurun_out/t1_parity.log
timeout 900 python -m pytest tests/test_gpu_runner.py tests/test_gpu_ddp.py -q -s -k "resume_on_device or world2_runner" > gpurun_out/t1_runner.log 2>&1; echo "rc $?" >> gpurun_out/t1_runner.log
tail -3 gpurun_out/t1_runner.log
timeout 900 python -m pytest tests/test_gpu_dataset.py -q -s > gpurun_out/t1_dataset.log 2>&1; echo "rc $?" >> gpurun_out/t1_dataset.log
tail -3 gpurun_out/t1_dataset.log
