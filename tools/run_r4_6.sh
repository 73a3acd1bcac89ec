#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 400 python tools/train_ab_route.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/ab_route2.txt; cat gpurun_out/r4/ab_route2.txt
for k in 1024 512; do P5_GEMM_RING128_MIN_K=$k timeout 200 python tools/gen_bench.py 20 10 2>&1 | grep -v amdgpu | sed "s/^/ring128_min_k=$k  /"; done | tee gpurun_out/r4/ab_gen_route.txt
timeout 600 python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu > gpurun_out/r4/bench_c5.json 2> gpurun_out/r4/bench_c5.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4/bench_c5.json').read().strip().splitlines()[-1])
print("C5", l["ms_per_step"], l["step_launches"])
for k in l["step_kernels"][:22]: print(k)
PY
bash tools/final_run_b.sh 2>&1 | tail -12
