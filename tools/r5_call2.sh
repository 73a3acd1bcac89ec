#!/bin/bash
# round 5, call 2: the gemm5 epilogue with every load issued before the first store (bench line), verified generation (timing + GPU tests)
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu --legs none --no-gen > gpurun_out/r5_bench1.log 2>&1
grep '^{' gpurun_out/r5_bench1.log | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.read())
kc = [k for k in l.get('step_kernels', []) if 'KC' in k['kernel'] and 'gemm5' in k['kernel']]
print('ms/step', l['ms_per_step'], 'roofline', l['roofline']['frac'], 'KC us/step', kc[0]['us_per_step'] if kc else None, [(g['grid'].split(' ')[-1], g['avg_us']) for g in (kc[0]['by_grid'] or [])] if kc else '')
" | tee gpurun_out/r5_call2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "verified or bf16_ranked_set or test_gemm or benchmark_shape or generate" 2>&1 | tail -15 | tee -a gpurun_out/r5_call2.txt
for mode in draft verified; do P5_GEN_MODE=$mode timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1; done | tee -a gpurun_out/r5_call2.txt
P5_GEN_MODE=verified P5_GEN_EXTRA=2 timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1 | tee -a gpurun_out/r5_call2.txt
P5_GEN_MODE=verified bash profiles/profile.sh r05_generate_verified_t5small_b20_k10 python tools/gen_bench.py 20 10 10
head -45 gpurun_out/r05_generate_verified_t5small_b20_k10.md | tee -a gpurun_out/r5_call2.txt
P5_VERIFY_SPLIT=0 P5_GEN_MODE=verified timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1 | sed 's/^/exact-fp32 verification: /' | tee -a gpurun_out/r5_call2.txt
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -12 | tee -a gpurun_out/r5_call2.txt
