#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
for m in verified draft; do for l in 2 3; do P5_GEN_MODE=$m timeout 200 python tools/gen_lanes_probe.py $l 20 2>&1 | tail -2; done; done | tee gpurun_out/r5_call10.txt
