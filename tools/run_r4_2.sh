#!/bin/bash
# round-4 call 2: reproducibility after the no-atomics change, whole GPU suite, A/B of the embedding-gradient path, default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/repro gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
for p in a b c; do timeout 300 python tools/diag_repro.py toy fix_$p 2 0.1 2>&1 | grep -v amdgpu.ids | tail -30; done > gpurun_out/repro/toy_fixed.txt
for p in a b c; do timeout 300 python tools/diag_repro.py c2 fix_$p 2 0.1 2>&1 | grep -v amdgpu.ids | tail -40; done > gpurun_out/repro/c2_fixed.txt
python tools/diag_repro.py cmp gpurun_out/repro/toy_fix_*.pt > gpurun_out/repro/cmp_fixed.txt 2>&1
python tools/diag_repro.py cmp gpurun_out/repro/c2_fix_*.pt >> gpurun_out/repro/cmp_fixed.txt 2>&1
rm -f gpurun_out/repro/*.pt
cat gpurun_out/repro/toy_fixed.txt gpurun_out/repro/c2_fixed.txt gpurun_out/repro/cmp_fixed.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r4/pytest2.log 2>&1
tail -25 gpurun_out/r4/pytest2.log
timeout 300 python tools/train_ab_det.py > gpurun_out/r4/ab_det.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4/ab_det.txt
timeout 600 python bench.py > gpurun_out/r4/bench2.json 2> gpurun_out/r4/bench2.err; tail -c 1500 gpurun_out/r4/bench2.json
