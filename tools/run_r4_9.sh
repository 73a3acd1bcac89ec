#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4
export PYTHONDONTWRITEBYTECODE=1
for rep in 1 2; do
for nv in 0 64 32; do P5_DEC_HEAD_NV=$nv timeout 200 python tools/gen_bench.py 20 10 2>&1 | grep -v amdgpu | sed "s/^/bf16 generation, head rows per workgroup (0 = auto 128): $nv  /"; done
done | tee gpurun_out/r4/ab_gen_head.txt
for nb in 0 16; do P5_GEN_DTYPE=fp32 P5_DEC_NB=$nb timeout 200 python tools/gen_bench.py 20 10 2>&1 | grep -v amdgpu | sed "s/^/fp32 generation, dec_nb=$nb  /"; done | tee gpurun_out/r4/ab_gen_fp32_nb.txt
for nv in 0 32 16; do P5_GEN_DTYPE=fp32 P5_DEC_HEAD_NV=$nv timeout 200 python tools/gen_bench.py 20 10 2>&1 | grep -v amdgpu | sed "s/^/fp32 generation, head rows per workgroup (0 = auto 64): $nv  /"; done | tee -a gpurun_out/r4/ab_gen_fp32_nb.txt
