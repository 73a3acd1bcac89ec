#!/bin/bash
# Builds tests/emu/libp5emu.so: the kernel sources of openp5_amd/csrc compiled for the HOST against the
# fiber emulator (hip_emu.h).  Test infrastructure only -- never loaded by the product package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
OBJ="$HERE/build"
mkdir -p "$OBJ"
FLAGS="-std=c++17 -O2 -g0 -fPIC -DP5_EMU -Wno-unused-value -Wno-vla-cxx-extension -I$ROOT/openp5_amd/csrc"
pids=()
for u in p5_lib p5_gemm_tu p5_attn_tu; do       # the translation units of the library, in parallel
  $CXX $FLAGS -include "$HERE/hip_emu.h" -x c++ -c "$ROOT/openp5_amd/csrc/$u.hip" -o "$OBJ/$u.o" & pids+=($!)
done
$CXX $FLAGS -include "$HERE/hip_emu.h" -c "$HERE/hip_emu.cpp" -o "$OBJ/hip_emu.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$CXX -shared -fPIC "$OBJ"/p5_lib.o "$OBJ"/p5_gemm_tu.o "$OBJ"/p5_attn_tu.o "$OBJ"/hip_emu.o -o "$HERE/libp5emu.so"
echo built "$HERE/libp5emu.so"
