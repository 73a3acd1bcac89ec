#!/bin/bash
# round 5, call 5: verified generation with the shared encoder / replay fast-forward / pipelined split GEMM; split GEMM probe; dataset gates (two datasets)
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call5.txt; : > $OUT
gb() { "$@" timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1; }
gb env P5_GEN_MODE=verified | tee -a $OUT
gb env P5_GEN_MODE=verified P5_SPLIT_PIPE=0 | sed 's/^/unpipelined split GEMM: /' | tee -a $OUT
gb env P5_GEN_MODE=verified P5_VERIFY_SPLIT=0 | sed 's/^/exact-fp32 MFMAs: /' | tee -a $OUT
gb env P5_GEN_MODE=draft | tee -a $OUT
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -9 | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generate or released or test_gemm" 2>&1 | tail -5 | tee -a $OUT
P5_GEN_MODE=verified bash profiles/profile.sh r05_generate_verified_t5small_b20_k10 python tools/gen_bench.py 20 10 10
head -40 gpurun_out/r05_generate_verified_t5small_b20_k10.md | tee -a $OUT
timeout 1500 python -m pytest tests/test_gpu_dataset.py -x -q -s 2>&1 | grep -v "^$" | tail -60 | tee -a $OUT
