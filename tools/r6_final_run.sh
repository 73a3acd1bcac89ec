#!/bin/bash
# Round-6 evidence run (one gpurun call): rocprofv3 tables of the training step and of generation (plain bf16 + verified), PMC traffic of the
# roofline GEMMs and of one decode step, default bench line (all legs), whole GPU suite.   gpurun --timeout 3600 -- 'bash tools/r6_final_run.sh'
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v "^W2026\|^E2026" | tail -6 > gpurun_out/r06_smoke.txt; cat gpurun_out/r06_smoke.txt
bash profiles/collect_pmc.sh > gpurun_out/r6_pmc.log 2>&1 || tail -5 gpurun_out/r6_pmc.log
for CTR in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/pmc_gen_$CTR; rm -rf $D
  P5_GEN_MODE=draft rocprofv3 --pmc $CTR --kernel-trace -d $D -o g -- python tools/gen_bench.py 20 5 10 > $D.log 2>&1 || tail -5 $D.log
done
python profiles/pmc_decode_step.py $(find gpurun_out/pmc_gen_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/pmc_gen_WRITE_SIZE -name "*_results.db" | head -1) \
  gpurun_out/pmc_decode_step.json > gpurun_out/r6_pmc_decode.log 2>&1 || tail -5 gpurun_out/r6_pmc_decode.log
rm -rf gpurun_out/pmc gpurun_out/pmc_gen_FETCH_SIZE gpurun_out/pmc_gen_WRITE_SIZE
bash profiles/profile.sh r06_train_t5small_b64 python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --no-pmc --legs none
P5_GEN_MODE=draft bash profiles/profile.sh r06_generate_t5small_b20_k10 python tools/gen_bench.py 20 10 10
P5_GEN_MODE=verified bash profiles/profile.sh r06_generate_verified_t5small_b20_k10 python tools/gen_bench.py 20 10 10
timeout 1500 python bench.py > gpurun_out/r6_bench_full.log 2>&1
grep '^{' gpurun_out/r6_bench_full.log | tail -1 > gpurun_out/r06_bench_line.json
python -c "
import json
l = json.load(open('gpurun_out/r06_bench_line.json'))
print('ms/step', l['ms_per_step'], 'value', l['value'], 'gen', l['generation']['items_per_s'], 'fallback', l['beam10_fallback_users_over_users'], 'trained', l['beam10_items_per_sec_trained_model'], l['beam10_fallback_users_over_users_trained_model'], 'plain', l['generation_plain_bf16']['items_per_s'])
r = l['roofline']; print('roofline', r['kernel'], r['frac'], r['frac_excl_dispatch'], r['us_per_step'], r['launches_per_step'], 'traffic', r['traffic'], r['traffic_stale'], 'mfma_busy', r['mfma_busy'])
print('alone', {k: v for k, v in r['alone'].items() if k in ('shape', 'avg_launch_us', 'achieved', 'traffic', 'algorithmic_bytes', 'mfma_busy')})
print('roofline_generation', {k: v for k, v in l['roofline_generation'].items() if k in ('frac', 'ms_per_step', 'traffic', 'traffic_stale', 'algorithmic_bytes')})
print('cpu_baseline', l.get('cpu_baseline'))
for k, v in (l.get('legs') or {}).items(): print('leg', k, {a: b for a, b in v.items() if not isinstance(b, (dict, list))} if isinstance(v, dict) else v)
for c in l['step_kernels'][:24]: print('   ', c['kernel'][:100], c['launches_per_step'], c['us_per_step'], c['tflops'])
"
timeout 2400 python -m pytest tests -q -m gpu -s --durations=15 > gpurun_out/r6_gpu_suite_full.log 2>&1
grep "\[dataset\]\|passed\|failed\|FAILED" gpurun_out/r6_gpu_suite_full.log | cut -c1-300 | tail -20
