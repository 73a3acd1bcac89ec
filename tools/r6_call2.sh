#!/bin/bash
# Round-6 call 2: gated-GELU epilogues + tile-wise AdamW on the hardware, option A/Bs of the C2 step (stagger, N=512 on the wave-specialised
# kernel, AdamW writing the copies, streaming stores), the bench line with the live PMC passes and the trained-model generation leg.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "gate or gated or adamw or reproducible or resume or trajectory or model_gated or converges" 2>&1 | grep -v "^W2026\|^E2026" | tail -14
timeout 600 python tools/train_ab_opts.py base= stag4=gemm5_stagger:4 stag8=gemm5_stagger:8 stag14=gemm5_stagger:14 n512g5=gemm_wide_min_tiles:128 adamflat=adam_tiles:0 2>&1 | grep "ms/step"
cp openp5_amd/libp5hip.so /tmp/libp5hip_product.so
cp tools/lab/ablate/libp5hip_nt.so openp5_amd/libp5hip.so && timeout 300 python tools/train_ab_opts.py nt_stores= nt_stag8=gemm5_stagger:8 2>&1 | grep "ms/step"
cp /tmp/libp5hip_product.so openp5_amd/libp5hip.so
timeout 900 python bench.py --legs none --no-cpu 2>gpurun_out/r6_call2_bench.err | grep '^{' > gpurun_out/r6_call2_bench.json
tail -3 gpurun_out/r6_call2_bench.err
python - <<'PY'
import json
l = json.load(open('gpurun_out/r6_call2_bench.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], 'fallback/users', l.get('beam10_fallback_users_over_users'))
print('trained', json.dumps(l['generation'].get('trained_model'))[:900])
r = l['roofline']
print('roofline', {k: r.get(k) for k in ('frac', 'frac_excl_dispatch', 'traffic', 'traffic_stale', 'mfma_busy', 'us_per_step')})
print('alone', {k: r['alone'].get(k) for k in ('avg_launch_us', 'achieved', 'traffic', 'traffic_read', 'traffic_write', 'algorithmic_bytes', 'mfma_busy', 'wait_frac_of_wave_cycles', 'issue_frac_of_wave_cycles')})
g = l.get('roofline_generation', {})
print('roofline_generation', {k: g.get(k) for k in ('frac', 'ms_per_step', 'traffic', 'traffic_stale', 'algorithmic_bytes')})
PY
} 2>&1 | tee gpurun_out/r6_call2.txt
