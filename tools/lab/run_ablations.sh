#!/bin/bash
# On the GPU box: the training step (C2) with each ablated library of tools/lab/build_ablations.sh swapped in for the product library
# (the box's copy of the tree is scratch).  Prints ms per step and the in-run profile row of the forward / data-gradient GEMMs.
cd "$(dirname "$0")/../.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
cp openp5_amd/libp5hip.so /tmp/libp5hip_product.so
one() {   # label
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1])
kc = [k for k in l.get('step_kernels', []) if 'KC' in k['kernel'] and 'gemm5' in k['kernel']]
print('$1', 'ms/step %.3f' % l['ms_per_step'], 'KC us/step %.1f' % (kc[0]['us_per_step'] if kc else -1), [(g['grid'].split(' ')[-1], g['avg_us']) for g in (kc[0]['by_grid'] or [])] if kc else '')
"
}
one product
for f in tools/lab/ablate/libp5hip_abl*.so; do
  cp "$f" openp5_amd/libp5hip.so
  one "$(basename $f .so | sed 's/libp5hip_//')"
done
cp /tmp/libp5hip_product.so openp5_amd/libp5hip.so
one product_again
