#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_ddp.py -q -x -m gpu -k "attention or model or reproducible or bf16_gradients or resume or staged or golden" > gpurun_out/r4/pytest7.log 2>&1; tail -3 gpurun_out/r4/pytest7.log
for i in 1 2; do timeout 300 python bench.py --legs none --no-cpu > gpurun_out/r4/bench7.json 2> gpurun_out/r4/bench7.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4/bench7.json').read().strip().splitlines()[-1])
print(l["ms_per_step"], l["beam10_items_per_sec"], l["step_launches"])
for k in l["step_kernels"][:14]: print(k)
PY
done
(cd tools/r03_snapshot && timeout 300 python bench.py --legs none --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-3 build  ms_per_step %.3f' % l['ms_per_step'])")
