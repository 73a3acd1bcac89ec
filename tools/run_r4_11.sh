#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm or model_bf16 or bf16_gradients_at or fused_loss or golden" > gpurun_out/r4/pytest11.log 2>&1; tail -2 gpurun_out/r4/pytest11.log
timeout 300 python bench.py --legs none --no-cpu > gpurun_out/r4/bench11.json 2> gpurun_out/r4/bench11.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4/bench11.json').read().strip().splitlines()[-1])
print(l["ms_per_step"], l["beam10_items_per_sec"], l["step_launches"])
for k in l["step_kernels"][:12]:
    print(k["kernel"][:100], k["launches_per_step"], k["us_per_step"], k["tflops"])
    for g in (k.get("by_grid") or []): print("      ", g)
PY
(cd tools/r03_snapshot && timeout 300 python bench.py --legs none --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-3 build  ms_per_step %.3f' % l['ms_per_step'])")
bash tools/run_r4_10.sh
