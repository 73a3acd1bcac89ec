#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash profiles/profile.sh r03a_train python bench.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
bash profiles/profile.sh r03a_train_serialized python tools/bench_noside.py --steps 10 --warmup 3 --no-cpu --no-gen --legs none
bash profiles/profile.sh r03a_gen python tools/gen_bench.py 20 5
