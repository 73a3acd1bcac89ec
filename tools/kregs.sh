#!/bin/bash
# Registers / spills / scratch of every kernel of a built translation unit, read from the object itself (no -save-temps rebuild):
#   bash tools/kregs.sh openp5_amd/build/p5_gemm_tu.o [name-pattern]
set -e
OBJ=${1:?object file}; PAT=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$OBJ"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$LLVM/llvm-readelf --notes $T/dev.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|vgpr_spill|private_segment_fixed|group_segment_fixed" | paste - - - - - - | grep -E "$PAT" | sed 's/  */ /g'
rm -rf $T
