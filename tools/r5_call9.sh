#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call9.txt; : > $OUT
gb() { "$@" timeout 120 python tools/gen_bench.py 20 20 ${KK:-10} 2>&1 | tail -1; }
gb env P5_GEN_MODE=verified | tee -a $OUT
gb env P5_GEN_MODE=draft | tee -a $OUT
KK=16 gb env P5_GEN_MODE=draft | tee -a $OUT
KK=20 gb env P5_GEN_MODE=draft | tee -a $OUT
KK=20 gb env P5_GEN_MODE=verified | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generate" 2>&1 | tail -3 | tee -a $OUT
P5_GEN_MODE=verified bash profiles/profile.sh r05b_generate_verified python tools/gen_bench.py 20 10 10
grep "beam_step\|verify_step\|split_kernel\|dec_score" gpurun_out/r05b_generate_verified.md | cut -c1-160 | tee -a $OUT
