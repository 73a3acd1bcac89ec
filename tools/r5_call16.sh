#!/bin/bash
# Round-5 call 16: the default bench line at HEAD (forced-prefix steps counted on the lane that ran the timed calls; C5 leg with the
# head-resident attention kernels) and the per-kernel table of a C5 step.   gpurun --timeout 900 -- 'bash tools/r5_call16.sh'
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r5_bench_full.log 2>&1
grep '^{' gpurun_out/r5_bench_full.log | tail -1 > gpurun_out/r05_bench_line.json
python -c "
import json
l = json.load(open('gpurun_out/r05_bench_line.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], 'plain', l['generation_plain_bf16']['items_per_s'])
print('roofline_generation', {k: l['roofline_generation'][k] for k in ('frac', 'ms_per_step', 'steps', 'forced_prefix_steps', 'achieved')})
print('legs', {k: (v.get('ms_per_step') or v.get('ms_per_batch')) for k, v in l['legs'].items()})
"
timeout 300 python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu 2>&1 | grep '^{' > gpurun_out/r05_c5_line.json
python - <<'PY'
import json
l = json.load(open('gpurun_out/r05_c5_line.json'))
out = ['# C5 per GPU (T5-large, B=64, L=512, T=10, bf16): per-kernel table of one training step from bench.py\'s in-run profiler',
       '# (`python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu`): %.1f ms/step, %d launches' % (l['ms_per_step'], l.get('step_launches', 0)),
       '', '| kernel | launches/step | us/step | TFLOP/s |', '|---|---|---|---|']
for c in l.get('step_kernels', []):
    out.append('| %s | %g | %.1f | %s |' % (c['kernel'], c['launches_per_step'], c['us_per_step'], c.get('tflops')))
open('gpurun_out/r05_c5_t5large_l512_step_kernels.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:14]))
PY
