#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_abi.py -q -x -m gpu > gpurun_out/r4/pytest3.log 2>&1; tail -5 gpurun_out/r4/pytest3.log
timeout 300 python tools/train_ab_route.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/ab_route.txt; cat gpurun_out/r4/ab_route.txt
bash profiles/profile.sh r4_gen_bf16 python tools/gen_bench.py 20 10 > gpurun_out/r4/prof_gen_bf16.log 2>&1; tail -2 gpurun_out/r4/prof_gen_bf16.log
P5_GEN_DTYPE=fp32 bash profiles/profile.sh r4_gen_fp32 python tools/gen_bench.py 20 10 > gpurun_out/r4/prof_gen_fp32.log 2>&1; tail -2 gpurun_out/r4/prof_gen_fp32.log
head -40 gpurun_out/r4_gen_fp32.md
timeout 900 python bench.py --legs configs > gpurun_out/r4/bench3.json 2> gpurun_out/r4/bench3.err; tail -c 600 gpurun_out/r4/bench3.json; tail -3 gpurun_out/r4/bench3.err
