#!/bin/bash
# Round-5 final call: smoke(), the default bench line, the C5 step table, the parity file of the GPU suite at HEAD.
#   gpurun --timeout 480 -- 'bash tools/r5_call27.sh'
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -3
timeout 300 python bench.py > gpurun_out/r5_bench_full.log 2>&1
grep '^{' gpurun_out/r5_bench_full.log | tail -1 > gpurun_out/r05_bench_line.json
python -c "
import json
l = json.load(open('gpurun_out/r05_bench_line.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], 'plain', l['generation_plain_bf16']['items_per_s'], 'loss', l['final_loss'])
print('roofline', {k: l['roofline'][k] for k in ('frac', 'achieved')}, 'roofline_generation', {k: l['roofline_generation'][k] for k in ('frac', 'ms_per_step', 'steps', 'forced_prefix_steps')})
print('legs', {k: (v.get('ms_per_step') or v.get('ms_per_batch')) for k, v in l['legs'].items()})
"
timeout 120 python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu 2>&1 | grep '^{' > gpurun_out/r05_c5_line.json
python - <<'PY'
import json
l = json.load(open('gpurun_out/r05_c5_line.json'))
out = ['# C5 per GPU (T5-large, B=64, L=512, T=10, bf16): per-kernel table of one training step from bench.py\'s in-run profiler',
       '# (`python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu`): %.1f ms/step, %d launches, final loss %s' % (l['ms_per_step'], l.get('step_launches', 0), l.get('final_loss')),
       '', '| kernel | launches/step | us/step | TFLOP/s |', '|---|---|---|---|']
for c in l.get('step_kernels', []):
    out.append('| %s | %g | %.1f | %s |' % (c['kernel'], c['launches_per_step'], c['us_per_step'], c.get('tflops')))
open('gpurun_out/r05_c5_t5large_l512_step_kernels.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:12]))
PY
timeout 420 python -m pytest tests/test_gpu_parity.py -q --durations=6 -k "not full_depth" > gpurun_out/r5_gpu_parity_final.log 2>&1
tail -12 gpurun_out/r5_gpu_parity_final.log
