#!/bin/bash
# The measurement calls of round 5 behind profiles/r05_call<N>_*.txt, one per case (each was one `gpurun -- bash tools/r5_calls.sh <N>`; the first
# call is tools/r5_first_call.sh, the round-end evidence run tools/r5_final_run.sh).  Kept as the provenance of those files, not as a workflow.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
case "$1" in
2)
# round 5, call 2: the gemm5 epilogue with every load issued before the first store (bench line), verified generation (timing + GPU tests)
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
  ;;
4)
# round 5, call 4: forced-prefix fast-forward + atomic-free decode step (generation timing, draft / verified / fp32), grid-barrier probe, GPU generation tests
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
  ;;
5)
# round 5, call 5: verified generation with the shared encoder / replay fast-forward / pipelined split GEMM; split GEMM probe; dataset gates (two datasets)
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
  ;;
6)
# round 5, call 6: split GEMM with raw loads (probe + verified timing), ML-1M-shaped dataset gate
OUT=gpurun_out/r5_call6.txt; : > $OUT
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -9 | tee -a $OUT
gb() { "$@" timeout 120 python tools/gen_bench.py 20 20 10 2>&1 | tail -1; }
gb env P5_GEN_MODE=verified | tee -a $OUT
gb env P5_GEN_MODE=verified P5_SPLIT_BIG_TILES=100000 | sed 's/^/64x64 split tiles only: /' | tee -a $OUT
gb env P5_GEN_MODE=verified P5_SPLIT_BIG_TILES=60 | sed 's/^/128x128 split tiles from 60: /' | tee -a $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "verified or test_gemm" 2>&1 | tail -3 | tee -a $OUT
timeout 1500 python -m pytest tests/test_gpu_dataset.py -x -q -s -k ml1m 2>&1 | grep -v "^$" | tail -30 | tee -a $OUT
  ;;
7)
# round 5, call 7: full GPU suite (incl. both dataset gates), default bench line, split GEMM probe rerun
OUT=gpurun_out/r5_call7.txt; : > $OUT
timeout 300 python tools/gemm_split_probe.py 2>&1 | tail -9 | tee -a $OUT
timeout 2400 python -m pytest tests -q -m gpu -s --durations=15 > gpurun_out/r5_gpu_suite_full.log 2>&1
grep "\[dataset\]\|\[verified\|passed\|failed\|FAILED\|slowest\|^[0-9.]*s call" gpurun_out/r5_gpu_suite_full.log | cut -c1-700 | tee -a $OUT
timeout 900 python bench.py > gpurun_out/r5_bench_full.log 2>&1
grep '^{' gpurun_out/r5_bench_full.log | tail -1 > gpurun_out/r5_bench_line.json
python - <<'PY' | tee -a $OUT
import json
l = json.load(open("gpurun_out/r5_bench_line.json"))
print("ms/step", l["ms_per_step"], "samples/s", l["value"], "roofline", {k: l["roofline"].get(k) for k in ("kernel", "frac", "frac_excl_dispatch", "bracket_floor_us", "us_per_step")})
print("generation", {k: l["generation"].get(k) for k in ("items_per_s", "ms_per_batch", "ms_per_batch_median_call", "verify_stats")})
print("plain bf16", {k: l["generation_plain_bf16"].get(k) for k in ("items_per_s", "ms_per_batch", "timing_ms")})
print("roofline_generation", l.get("roofline_generation"))
for k, v in (l.get("legs") or {}).items(): print("leg", k, v)
print("cpu", l.get("cpu_baseline"), l.get("cpu_baseline_generation"))
PY
  ;;
9)
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
  ;;
10)
for m in verified draft; do for l in 2 3; do P5_GEN_MODE=$m timeout 200 python tools/gen_lanes_probe.py $l 20 2>&1 | tail -2; done; done | tee gpurun_out/r5_call10.txt
  ;;
11)
OUT=gpurun_out/r5_call11.txt; : > $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py tests/test_gpu_ddp.py -x -q -k "lanes or runner or filtered or world2 or two_ranks or resume" 2>&1 | tail -5 | tee -a $OUT
timeout 600 python bench.py --no-cpu --legs none --steps 5 --warmup 2 > gpurun_out/r5_bench_lanes.log 2>&1
grep '^{' gpurun_out/r5_bench_lanes.log | tail -1 | python -c "
import sys, json
l = json.loads(sys.stdin.read())
g = l['generation']; d = l['generation_plain_bf16']
print('verified', {k: g.get(k) for k in ('items_per_s', 'ms_per_batch', 'ms_per_batch_median_call', 'lanes', 'verify_stats')})
print('after noise', g.get('after_noise_training'))
print('plain', {k: d.get(k) for k in ('items_per_s', 'ms_per_batch', 'ms_per_batch_median_call')})
" | tee -a $OUT
tail -3 gpurun_out/r5_bench_lanes.log | cut -c1-300
  ;;
12)
for m in verified draft; do for l in 2 3 4; do P5_GEN_LANES=$l P5_GEN_MODE=$m timeout 200 python tools/gen_bench.py 20 10 10 2>&1 | grep lanes: | sed "s/^/$m /"; done; done | tee gpurun_out/r5_call12.txt
  ;;
*) echo "usage: $0 <2|4|5|6|7|9|10|11|12>"; exit 1;;
esac
