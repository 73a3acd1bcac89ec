#!/bin/bash
# Lab builds of the PRODUCT library with parts of the forward / data-gradient instance of p5_gemm5 removed (template parameter ABL,
# p5_gemm5.h), for timing INSIDE the training step (round 3's lab timed the ablated kernels alone, on warm operands; in the step the
# epilogue writes cold output and every workgroup is in the same phase).  Runs here (hipcc cross-compiles, ~2 min per variant):
#     bash tools/lab/build_ablations.sh            ->  tools/lab/ablate/libp5hip_abl<bits>.so      (git-ignored, travels with gpurun)
# then on the GPU box:  bash tools/lab/run_ablations.sh
# bits: 8 = no epilogue at all, 16 = epilogue arithmetic without the stores, 32 = no dropout hash, 64 = no residual / saved-hidden read,
#       2 = no operand copies after the prologue (K loop on stale LDS), 1 = no MFMAs.  Results of an ablated build are garbage: timing only.
set -e
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CS=openp5_amd/csrc
OUT=tools/lab/ablate
mkdir -p $OUT openp5_amd/build
for u in p5_lib p5_attn_tu; do          # the other translation units are the product's own objects
  [ -f openp5_amd/build/$u.o ] || { echo "build the product first (python -c 'import __graft_entry__ as g; g.build()')"; exit 1; }
done
for ABL in ${ABLS:-8 16 32 64 96 2 1}; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DP5_GEMM5_ABL=$ABL -I $CS -c $CS/p5_gemm_tu.hip -o $OUT/p5_gemm_tu_abl$ABL.o &
done
wait
for ABL in ${ABLS:-8 16 32 64 96 2 1}; do
  $HIPCC --offload-arch=gfx950 -shared -fPIC openp5_amd/build/p5_lib.o $OUT/p5_gemm_tu_abl$ABL.o openp5_amd/build/p5_attn_tu.o -o $OUT/libp5hip_abl$ABL.so
  rm -f $OUT/p5_gemm_tu_abl$ABL.o
  echo "built $OUT/libp5hip_abl$ABL.so"
done
