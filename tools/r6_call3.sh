#!/bin/bash
# Round-6 call 3: logit-free cross-entropy + the reworked tile-wise AdamW (64 x 256 tiles) on the hardware: parity tests, in-process A/Bs, bench line.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "logit_free or adamw or gate or gated or reproducible or resume or trajectory or benchmark_shape or fused_loss or dropout_on or model_t5_small" 2>&1 | grep -v "^W2026\|^E2026" | tail -16
timeout 600 python tools/train_ab_opts.py base= ce_mat=ce_free:0 adamflat=adam_tiles:0 both_old=ce_free:0,adam_tiles:0 2>&1 | grep "ms/step"
timeout 900 python bench.py --legs none --no-cpu 2>gpurun_out/r6_call3_bench.err | grep '^{' > gpurun_out/r6_call3_bench.json
tail -3 gpurun_out/r6_call3_bench.err
python - <<'PY'
import json
l = json.load(open('gpurun_out/r6_call3_bench.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], 'fallback/users', l.get('beam10_fallback_users_over_users'))
t = l['generation'].get('trained_model', {})
print('trained', {k: v for k, v in t.items() if k not in ('note', 'verify_stats')})
r = l['roofline']
print('roofline', {k: r.get(k) for k in ('frac', 'frac_excl_dispatch', 'traffic', 'traffic_stale', 'mfma_busy', 'us_per_step')})
for c in l.get('step_kernels', [])[:30]: print('   ', c['kernel'][:100], c['launches_per_step'], round(c['us_per_step']), c.get('tflops'))
PY
} 2>&1 | tee gpurun_out/r6_call3.txt
