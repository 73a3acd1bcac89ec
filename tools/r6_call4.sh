#!/bin/bash
# Round-6 call 4: the tr-read / partial-wait hazard probe, option A/B (N = 512, K = 512 GEMMs on the ring kernel), the whole GPU suite at HEAD.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
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
