#!/bin/bash
# Round-end evidence run, gpurun call B: kernel-trace profiles of training and generation (per kernel and per launch grid), and the same-box
# A/B of this build against the round-3 build (tools/r03_snapshot: the round-3 tree with its own libp5hip.so; not tracked).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_dataset.py -q -x -m gpu -s > gpurun_out/final_pytest_dataset.log 2>&1; grep "dataset\]\|passed\|failed" gpurun_out/final_pytest_dataset.log | cut -c1-300 | tail -12
bash profiles/profile.sh final_train python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
bash profiles/profile.sh final_gen python tools/gen_bench.py 20 5
# the fp32 (list-identical) generation mode with narrower column tiles of the norm-fused projections (two workgroups per CU)
for nb in 0 16; do P5_GEN_DTYPE=fp32 P5_DEC_NB=$nb timeout 200 python tools/gen_bench.py 20 10 2>&1 | grep -v amdgpu | sed "s/^/fp32 generation, dec_nb=$nb  /"; done | tee gpurun_out/ab_gen_fp32_nb.txt
if [ -d tools/r03_snapshot ]; then
  : > gpurun_out/ab_r03_r04.txt
  for rep in 1 2 3; do
    (cd tools/r03_snapshot && timeout 300 python bench.py --legs none --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-3 build  ms_per_step %.3f  samples/s %.0f  gen ms/batch %.3f  items/s %.0f' % (l['ms_per_step'], l['value'], l['generation']['ms_per_batch'], l['beam10_items_per_sec']))") >> gpurun_out/ab_r03_r04.txt
    (timeout 300 python bench.py --legs none --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-4 build  ms_per_step %.3f  samples/s %.0f  gen ms/batch %.3f  items/s %.0f' % (l['ms_per_step'], l['value'], l['generation']['ms_per_batch'], l['beam10_items_per_sec']))") >> gpurun_out/ab_r03_r04.txt
  done
  cat gpurun_out/ab_r03_r04.txt
fi
