#!/bin/bash
# Round-5 evidence run (one gpurun call): default bench line, rocprofv3 tables of the training step and of generation (plain bf16 + verified),
# PMC traffic of the roofline GEMMs and of one decode step, grid-barrier probe.   gpurun --timeout 2400 -- 'bash tools/r5_final_run.sh'
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
bash profiles/collect_pmc.sh > gpurun_out/r5_pmc.log 2>&1 || tail -5 gpurun_out/r5_pmc.log
for CTR in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/pmc_gen_$CTR; rm -rf $D
  P5_GEN_MODE=draft rocprofv3 --pmc $CTR --kernel-trace -d $D -o g -- python tools/gen_bench.py 20 5 10 > $D.log 2>&1 || tail -5 $D.log
done
python profiles/pmc_decode_step.py $(find gpurun_out/pmc_gen_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/pmc_gen_WRITE_SIZE -name "*_results.db" | head -1) \
  gpurun_out/pmc_decode_step.json > gpurun_out/r5_pmc_decode.log 2>&1 || tail -5 gpurun_out/r5_pmc_decode.log
cp gpurun_out/pmc_decode_step.json profiles/pmc_decode_step.json 2>/dev/null
rm -rf gpurun_out/pmc gpurun_out/pmc_gen_FETCH_SIZE gpurun_out/pmc_gen_WRITE_SIZE
bash profiles/profile.sh r05_train_t5small_b64 python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
P5_GEN_MODE=draft bash profiles/profile.sh r05_generate_t5small_b20_k10 python tools/gen_bench.py 20 10 10
P5_GEN_MODE=verified bash profiles/profile.sh r05_generate_verified_t5small_b20_k10 python tools/gen_bench.py 20 10 10
timeout 60 tools/probe/grid_barrier.bin > gpurun_out/r05_grid_barrier_probe.txt 2>&1
timeout 900 python bench.py > gpurun_out/r5_bench_full.log 2>&1
grep '^{' gpurun_out/r5_bench_full.log | tail -1 > gpurun_out/r05_bench_line.json
python -c "
import json
l = json.load(open('gpurun_out/r05_bench_line.json'))
print('ms/step', l['ms_per_step'], 'gen', l['generation']['items_per_s'], l['generation']['verify_stats'], 'after noise', l['generation'].get('after_noise_training'), 'plain', l['generation_plain_bf16']['items_per_s'])
print('roofline_generation', l.get('roofline_generation'))
"
timeout 2400 python -m pytest tests -q -m gpu -s --durations=15 > gpurun_out/r5_gpu_suite_full.log 2>&1
grep "\[dataset\]\|\[verified\|passed\|failed\|FAILED" gpurun_out/r5_gpu_suite_full.log | cut -c1-400 | tail -40
