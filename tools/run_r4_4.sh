#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_ddp.py -q -x -m gpu -k "attention or model or reproducible or bf16_gradients or resume or staged or stored or golden or rmsnorm" > gpurun_out/r4/pytest4.log 2>&1; tail -6 gpurun_out/r4/pytest4.log
timeout 400 python tools/train_ab_route.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/ab_route.txt; cat gpurun_out/r4/ab_route.txt
timeout 300 python bench.py --legs none --no-cpu > gpurun_out/r4/bench4.json 2> gpurun_out/r4/bench4.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4/bench4.json').read().strip().splitlines()[-1])
print(l["ms_per_step"], l["beam10_items_per_sec"], l["step_launches"])
for k in l["step_kernels"][:30]: print(k)
PY
timeout 600 python tools/gate_explore.py chain 30 1e-3 2 > gpurun_out/r4/gate_chain30.txt 2>&1; grep -v amdgpu gpurun_out/r4/gate_chain30.txt | tail -8
timeout 600 python tools/gate_explore.py none 30 1e-3 2 > gpurun_out/r4/gate_none30.txt 2>&1; grep -v amdgpu gpurun_out/r4/gate_none30.txt | tail -8
