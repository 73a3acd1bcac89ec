#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/t9_bench.txt
run() {
  local script=$1; shift
  echo "== $script $*" >> gpurun_out/t9_bench.txt
  env "$@" timeout 300 python $script --steps 20 --warmup 5 --no-cpu --legs none 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'], d['generation']['ms_per_batch'], d['generation']['timing_ms']['encode_ms'], d['generation']['timing_ms']['decode_ms'])" >> gpurun_out/t9_bench.txt 2>&1
}
run bench.py P5_NORM_FUSE=1
run bench.py P5_NORM_FUSE=0
run bench.py P5_NORM_FUSE=1
run bench.py P5_NORM_FUSE=0
cat gpurun_out/t9_bench.txt
bash profiles/profile.sh r03b_train_nf1 python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
