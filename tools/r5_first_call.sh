#!/bin/bash
# Round 5, first GPU call -- measurement before any change:
#   here (CPU, ~8 min):   python -c "import __graft_entry__ as g; g.build()" && bash tools/lab/build_ablations.sh
#   then:                 gpurun --timeout 1500 -- 'bash tools/r5_first_call.sh'
# On the box: the default bench line (is the box comparable with round 4's 4.26 ms?), then the training step with each ablated library.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu --legs none > gpurun_out/r5_bench0.log 2>&1
grep '^{' gpurun_out/r5_bench0.log | tail -1 | python -c "import sys, json; l = json.loads(sys.stdin.read()); print('ms/step', l['ms_per_step'], 'roofline', l['roofline']['frac'], 'launches', l.get('step_launches'))"
bash tools/lab/run_ablations.sh 2>&1 | tee gpurun_out/r5_ablations_in_step.txt
# generation: what a wider bf16 beam and the fp32 engine cost on this box (round-5 task 1 planning)
for k in 10 16; do timeout 120 python tools/gen_bench.py 20 20 $k 2>&1 | tail -1; done | tee gpurun_out/r5_gen_widths.txt
P5_GEN_DTYPE=fp32 timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1 | tee -a gpurun_out/r5_gen_widths.txt
P5_GEN_DTYPE=fp32 timeout 120 python tools/gen_bench.py 20 20 16 2>&1 | tail -1 | tee -a gpurun_out/r5_gen_widths.txt
