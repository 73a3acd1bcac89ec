#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): full GPU suite, the default bench line, kernel-trace profiles of
# training and generation (per kernel and per launch grid), PMC traffic of the roofline kernels.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/final_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest.log
tail -4 gpurun_out/final_pytest.log
bash profiles/collect_pmc.sh > gpurun_out/final_pmc.log 2>&1; tail -3 gpurun_out/final_pmc.log
timeout 600 python bench.py > gpurun_out/final_bench.log 2>&1; grep '^{' gpurun_out/final_bench.log | tail -1 > gpurun_out/final_bench.json; tail -c 300 gpurun_out/final_bench.json
bash profiles/profile.sh final_train python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
bash profiles/profile.sh final_gen python tools/gen_bench.py 20 5
bash profiles/profile.sh final_train_serialized python tools/bench_noside.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
python tools/attn_bench.py > gpurun_out/final_attn_bench.txt 2>&1; tail -4 gpurun_out/final_attn_bench.txt
