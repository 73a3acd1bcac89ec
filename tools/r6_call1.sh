#!/bin/bash
# Round-6 call 1: the round's new parity tests on the hardware (verified generation at K = 20 / 22 / 23 on T5-base dims, split range guard,
# AdamW against the published-4.26 fixture, ABI v4), the bench line at HEAD, and the side-stream A/B of the C2 step.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "verified or adamw or abi or version or range_guard or generate" 2>&1 | grep -v "^W2026\|^E2026" | tail -25
for a in "0 1" "1 1" "0 1" "1 1" "1 3"; do timeout 200 python tools/train_ab_side.py $a 2>&1 | grep side_stream; done
timeout 600 python bench.py --legs none --no-cpu 2>gpurun_out/r6_call1_bench.err | grep '^{' > gpurun_out/r6_call1_bench.json
python - <<'PY'
import json
l = json.load(open('gpurun_out/r6_call1_bench.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], l['generation']['verify_stats'], 'plain', l['generation_plain_bf16']['items_per_s'])
print('roofline', {k: l['roofline'][k] for k in ('kernel', 'frac', 'frac_excl_dispatch', 'us_per_step', 'launches_per_step')})
for c in l.get('step_kernels', [])[:24]: print('   ', c['kernel'][:100], c['launches_per_step'], round(c['us_per_step']), c.get('tflops'))
PY
} 2>&1 | tee gpurun_out/r6_call1.txt
