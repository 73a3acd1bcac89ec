#!/bin/bash
# The round-6 measurement calls, one case per `profiles/r06_call<N>_*.txt`:   gpurun --timeout 2400 -- bash tools/r6_calls.sh <N>
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
case "$1" in
1)
  # Round-6 call 1: the round's new parity tests on the hardware (verified generation at K = 20 / 22 / 23 on T5-base dims, split range guard,
# AdamW against the published-4.26 fixture, ABI v4), the bench line at HEAD, and the side-stream A/B of the C2 step.
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
;;
2)
  # Round-6 call 2: gated-GELU epilogues + tile-wise AdamW on the hardware, option A/Bs of the C2 step (stagger, N=512 on the wave-specialised
# kernel, AdamW writing the copies, streaming stores), the bench line with the live PMC passes and the trained-model generation leg.
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
;;
3)
  # Round-6 call 3: logit-free cross-entropy + the reworked tile-wise AdamW (64 x 256 tiles) on the hardware: parity tests, in-process A/Bs, bench line.
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
;;
4)
  # Round-6 call 4: the tr-read / partial-wait hazard probe, option A/B (N = 512, K = 512 GEMMs on the ring kernel), the whole GPU suite at HEAD.
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 300 python tools/probe/tr_wait_probe.py 2>&1 | grep -v "^W2026\|^E2026" | tail -12
timeout 400 python tools/train_ab_opts.py base= ring512=gemm_ring128_min_k:512 2>&1 | grep "ms/step"
timeout 300 python bench.py --legs none --no-cpu --no-gen --no-pmc 2>/dev/null | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.read()); print('ms/step', l['ms_per_step'])
for c in l.get('step_kernels', []): 
    if 'embed' in c['kernel'] or 'adamw' in c['kernel'] or 'sumsq' in c['kernel']: print('   ', c['kernel'][:80], c['launches_per_step'], round(c['us_per_step'], 1))
"
timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | grep -v "^W2026\|^E2026" | tail -30
} 2>&1 | tee gpurun_out/r6_call4.txt
;;
5)
  # Round-6 call 5: wave-specialised 128x128 instance for the N = d_model GEMMs (parity on the hardware, in-step A/B incl. K = 512), workgroup count of the norm backward.
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wave_specialised or bf16_gradients_at_benchmark or test_model_bf16 or reproducible" 2>&1 | grep -v "^W2026\|^E2026" | tail -6
timeout 600 python tools/train_ab6.py old=gemm_ws128:0,norm_bwd_blocks:1024,gemm_ws128_min_k:512 ws128=gemm_ws128:1 ws128k1024=gemm_ws128:1,gemm_ws128_min_k:1024 nb512=norm_bwd_blocks:512 nb256=norm_bwd_blocks:256 both=gemm_ws128:1,norm_bwd_blocks:512 2>&1 | grep "ms/step"
timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
P5_NORM_BWD_BLOCKS=512 timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
P5_NORM_BWD_BLOCKS=256 timeout 200 python tools/elem_bench.py 2>&1 | grep rmsnorm_bwd
} 2>&1 | tee gpurun_out/r6_call5.txt
;;
6)
  # Round-6 call 6: T5LayerNorm backward in the data-gradient GEMM epilogues -- parity on the hardware (op level, C2 gradients, reproducibility), in-step A/B.
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k "norm_backward or row_sums or wave_specialised or bf16_gradients_at_benchmark or test_model_bf16 or reproducible or gated or resume" 2>&1 | grep -v "^W2026\|^E2026" | tail -25
timeout 600 python tools/train_ab6.py --show gemm5,gemm2,p5_gemm_kernel,rmsnorm_bwd,attn_bwd_fused,reduce_rows old=norm_bwd_fuse:0,gemm_ws128:0 ws128=gemm_ws128:1 fuse=norm_bwd_fuse:1,gemm_ws128:1 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call6.txt
;;
7)
  # Round-6 call 7: norm-backward epilogue with n written by the loader waves -- parity (op level, C2 gradients), in-step A/B incl. side-stream settings.
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "norm_backward or bf16_gradients_at_benchmark or reproducible" 2>&1 | grep -v "^W2026\|^E2026" | tail -4
timeout 900 python tools/train_ab6.py --show gemm5,rmsnorm_bwd,attn_bwd_fused old=norm_bwd_fuse:0,gemm_ws128:0,wgrad_side:1 ws128=gemm_ws128:1 fuse=norm_bwd_fuse:1,gemm_ws128:1 fuse_s3=norm_bwd_fuse:1,gemm_ws128:1,wgrad_side:3 nofuse_s3=norm_bwd_fuse:0,gemm_ws128:1,wgrad_side:3 fuse_s0=norm_bwd_fuse:1,gemm_ws128:1,wgrad_side:0 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call7.txt
;;
8)
  # Round-6 call 8: timeline of the C2 step (idle time, overlap, gaps) from a rocprofv3 kernel trace; decode-step skinny GEMM with an XCD-aware unit order (lab patch, not kept: no change in FETCH_SIZE -- the column-tile counts are multiples of 8, so the row tiles of a column tile already share an XCD).
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
OUT=gpurun_out/prof_tl; rm -rf $OUT; mkdir -p $OUT
( rocprofv3 --kernel-trace -d $OUT -o t -- python bench.py --legs none --no-cpu --no-gen --no-pmc --steps 12 --warmup 4 ) > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
DB=$(find $OUT -name "*_results.db" | head -1)
python profiles/timeline_rocpd.py "$DB" gpurun_out/r6_timeline.txt
rm -rf $OUT/*.db
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generate or skinny or decode" 2>&1 | grep -v "^W2026\|^E2026" | tail -4
for x in 1 0; do
  P5_DEC_XCD=$x timeout 300 python bench.py --legs none --no-cpu --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.read()); g = l['roofline_generation']
print('dec_xcd=$x', 'ms/step', l['ms_per_step'], 'gen items/s', l['beam10_items_per_sec'], 'decode step ms', g['ms_per_step'], 'traffic MB', round((g.get('traffic') or 0) / 1e6, 1), 'stale', g.get('traffic_stale'), 'plain', l['generation_plain_bf16'].get('items_per_s'))
"
done
} 2>&1 | tee gpurun_out/r6_call8.txt
;;
10)
  # Round-6 call 10: norm-backward epilogue beyond the one-tile-per-CU case (option norm_bwd_fuse 2) at C3 (T5-base) and C5 (T5-large, L = 512).
{
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 600 python tools/leg_ab.py c3 fuse1=norm_bwd_fuse:1 fuse2=norm_bwd_fuse:2 fuse0=norm_bwd_fuse:0 2>&1 | grep "ms/step"
timeout 900 python tools/leg_ab.py c5 fuse1=norm_bwd_fuse:1 fuse2=norm_bwd_fuse:2 2>&1 | grep "ms/step"
} 2>&1 | tee gpurun_out/r6_call10.txt
;;
11)
  # Round-6 call 11: per-kernel tables of the C3 (T5-base) and C5 (T5-large, L = 512) steps.
{
timeout 600 python tools/leg_profile.py c3 2>&1 | grep -v "^W2026\|^E2026"
timeout 900 python tools/leg_profile.py c5 2>&1 | grep -v "^W2026\|^E2026"
} 2>&1 | tee gpurun_out/r6_call11.txt
;;
*) echo "usage: r6_calls.sh 1|2|3|4|5|6|7|8|10|11"; exit 2 ;;
esac
