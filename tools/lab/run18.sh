#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/lab/gemm_lab lab4 > gpurun_out/lab4b.txt 2>&1
grep -v "M=8000" gpurun_out/lab4b.txt | head -45
