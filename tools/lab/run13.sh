#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t13_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t13_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --no-gen --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/t13_bench.txt 2>&1
}
run bench.py A=1
run bench.py P5_GEMM_SMALL_RING=0
run bench.py P5_GEMM_RING32=0
run bench.py P5_GEMM_SMALL_RING=0 P5_GEMM_RING32=0
run bench.py P5_DGRAD_T=0
run bench.py P5_ATTN_FUSED=0
run bench.py P5_ATTN_FWD_WG=0
run bench.py P5_GEMM_XCD_RECT=0
run bench.py A=1
run bench.py P5_GEMM_SMALL_RING=0 P5_GEMM_RING32=0
cat gpurun_out/t13_bench.txt
