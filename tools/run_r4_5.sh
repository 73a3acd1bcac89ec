#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_dataset.py tests/test_gpu_parity.py -q -x -m gpu -s -k "dataset or attention_short or longer_than_64 or test_attention" > gpurun_out/r4/pytest5.log 2>&1; grep "dataset\]\|passed\|failed\|Error\|assert" gpurun_out/r4/pytest5.log | cut -c1-400 | tail -30
timeout 300 python bench.py --legs none --no-cpu > gpurun_out/r4/bench5.json 2> gpurun_out/r4/bench5.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4/bench5.json').read().strip().splitlines()[-1])
print(l["ms_per_step"], l["beam10_items_per_sec"], l["step_launches"])
for k in l["step_kernels"][:32]: print(k)
PY
