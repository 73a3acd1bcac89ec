#!/bin/bash
# round 5, call 4: forced-prefix fast-forward + atomic-free decode step (generation timing, draft / verified / fp32), grid-barrier probe, GPU generation tests
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
OUT=gpurun_out/r5_call4.txt; : > $OUT
timeout 60 tools/probe/grid_barrier.bin 2>&1 | tee -a $OUT
gb() { "$@" timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1; }
echo "--- plain bf16 search (draft), K=10" | tee -a $OUT
gb env P5_GEN_MODE=draft | tee -a $OUT
gb env P5_GEN_MODE=draft P5_GEN_FF=0 | sed 's/^/no fast-forward: /' | tee -a $OUT
gb env P5_GEN_MODE=draft P5_DEC_ATOMIC=1 | sed 's/^/atomic residual updates: /' | tee -a $OUT
gb env P5_GEN_MODE=draft P5_GEN_FF=0 P5_DEC_ATOMIC=1 | sed 's/^/round-4 configuration: /' | tee -a $OUT
echo "--- verified" | tee -a $OUT
gb env P5_GEN_MODE=verified | tee -a $OUT
gb env P5_GEN_MODE=verified P5_GEN_EXTRA=4 | tee -a $OUT
echo "--- fp32 engine" | tee -a $OUT
gb env P5_GEN_DTYPE=fp32 | tee -a $OUT
gb env P5_GEN_DTYPE=fp32 P5_GEN_FF=0 P5_DEC_ATOMIC=1 | sed 's/^/round-4 configuration: /' | tee -a $OUT
echo "--- 64 users per batch" | tee -a $OUT
for m in draft verified; do P5_GEN_MODE=$m timeout 120 python tools/gen_bench.py 64 10 10 2>&1 | tail -1 | tee -a $OUT; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "generate or skinny or decode or cross_attn" 2>&1 | tail -6 | tee -a $OUT
P5_GEN_MODE=draft bash profiles/profile.sh r05_generate_t5small_b20_k10 python tools/gen_bench.py 20 10 10
head -30 gpurun_out/r05_generate_t5small_b20_k10.md | tee -a $OUT
