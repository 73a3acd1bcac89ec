#!/bin/bash
# round-4 call 1: MFMA ceiling of the box + run-to-run reproducibility of the toy training and of the C2 step, fresh processes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/repro
export PYTHONDONTWRITEBYTECODE=1
tools/lab/mfma_peak.bin > gpurun_out/mfma_peak.txt 2>&1
tail -12 gpurun_out/mfma_peak.txt
for p in a b c; do timeout 300 python tools/diag_repro.py toy def_$p 2 0.1 2>&1 | grep -v Warning | tail -30; done > gpurun_out/repro/toy_default.txt
for p in a b; do timeout 300 python tools/diag_repro.py toy nodrop_$p 2 0.0 2>&1 | tail -30; done > gpurun_out/repro/toy_nodrop.txt
P5_GRAD_STORE_FIRST=0 P5_NORM_FUSE=0 P5_WGRAD_GROUP=0 timeout 300 python tools/diag_repro.py toy r2paths_a 2 0.1 > gpurun_out/repro/toy_r2paths.txt 2>&1
for p in a b c; do timeout 300 python tools/diag_repro.py c2 def_$p 2 0.1 2>&1 | tail -40; done > gpurun_out/repro/c2_default.txt
python tools/diag_repro.py cmp gpurun_out/repro/toy_def_*.pt > gpurun_out/repro/cmp_toy.txt 2>&1
python tools/diag_repro.py cmp gpurun_out/repro/toy_nodrop_*.pt >> gpurun_out/repro/cmp_toy.txt 2>&1
python tools/diag_repro.py cmp gpurun_out/repro/c2_def_*.pt > gpurun_out/repro/cmp_c2.txt 2>&1
rm -f gpurun_out/repro/*.pt
cat gpurun_out/repro/toy_default.txt gpurun_out/repro/toy_nodrop.txt gpurun_out/repro/toy_r2paths.txt gpurun_out/repro/c2_default.txt gpurun_out/repro/cmp_toy.txt gpurun_out/repro/cmp_c2.txt | tail -120
