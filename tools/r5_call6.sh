#!/bin/bash
# round 5, call 6: split GEMM with raw loads (probe + verified timing), ML-1M-shaped dataset gate
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call6.txt; : > $OUT
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -9 | tee -a $OUT
gb() { "$@" timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1; }
gb env P5_GEN_MODE=verified | tee -a $OUT
gb env P5_GEN_MODE=verified P5_SPLIT_BIG_TILES=100000 | sed 's/^/64x64 split tiles only: /' | tee -a $OUT
gb env P5_GEN_MODE=verified P5_SPLIT_BIG_TILES=60 | sed 's/^/128x128 split tiles from 60: /' | tee -a $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "verified or test_gemm" 2>&1 | tail -3 | tee -a $OUT
timeout 1500 python -m pytest tests/test_gpu_dataset.py -x -q -s -k ml1m 2>&1 | grep -v "^$" | tail -30 | tee -a $OUT
