#!/bin/bash
# Round-5 call 23: the long-sequence forward that also stores its dropout masks, final form (masks collected by selects, stored one block
# later): equality with the plain forward over repeated runs, the parity tests, the C5 step with / without the masks.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "storing_masks" 2>&1 | tail -2      # (was tools/tmp/dbg_keep4.py, now tests/cases.py::attn_keep_masks_forward_case)
timeout 400 python -m pytest tests/test_gpu_parity.py -q -k "head_resident or keep_masks or attention or L512" 2>&1 | tail -4
for BITS in 1 0; do
  echo "C5 step, P5_ATTN_KEEP_BITS=$BITS"
  P5_ATTN_KEEP_BITS=$BITS timeout 300 python bench.py --backbone t5-large --seq-len 512 --tgt-len 10 --steps 3 --warmup 1 --legs none --no-gen --no-cpu 2>&1 | grep '^{' | python -c "
import sys, json
l = json.loads(sys.stdin.read()); print('ms/step', l['ms_per_step'], 'samples/s', l['value'], 'final_loss', l.get('final_loss'))
for c in l.get('step_kernels', []): 
    if 'attn' in c['kernel']: print('   ', c['kernel'][:70], c['launches_per_step'], round(c['us_per_step']), c.get('tflops'))
"
done
} 2>&1 | tee gpurun_out/r5_call23.txt
