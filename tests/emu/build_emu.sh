#!/bin/bash
# Builds tests/emu/libp5emu.so: the kernel sources of openp5_amd/csrc compiled for the HOST against the
# fiber emulator (hip_emu.h).  Test infrastructure only -- never loaded by the product package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -std=c++17 -O2 -g0 -fPIC -shared -DP5_EMU -Wno-unused-value -Wno-vla-cxx-extension \
  -include "$HERE/hip_emu.h" -x c++ "$ROOT/openp5_amd/csrc/p5_lib.hip" "$HERE/hip_emu.cpp" \
  -I"$ROOT/openp5_amd/csrc" -o "$HERE/libp5emu.so"
echo built "$HERE/libp5emu.so"
