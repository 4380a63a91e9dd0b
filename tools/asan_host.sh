#!/bin/bash
# Host-side AddressSanitizer + UBSan run (CPU only; GPU sanitizers are not available on this pool): builds the library with
# the host code instrumented into /tmp/relp_asan and runs the CPU tests that exercise the C++ host (parser, standardisation,
# presolve, graph providers, big rationals, ABI) against it through RELP_AMD_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/relp_asan
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
cd $ROOT/relp_amd/csrc
rm -f $OUT/*.o
# (the sources of the Makefile: every .hip and .cpp of the library)
SRC=$(sed -n 's/^SRC := //p' Makefile)
for f in $SRC; do case $f in *.hip) /opt/rocm/bin/hipcc $FLAGS -c $f -o $OUT/${f%.hip}.o & ;; *.cpp) /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o $OUT/${f%.cpp}.o & ;; esac; done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $OUT/librelp_amd.so $OUT/*.o
cd $ROOT
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 RELP_AMD_LIB=$OUT/librelp_amd.so \
  python -m pytest tests/test_host_model.py tests/test_host_general_form.py tests/test_host_presolve.py tests/test_abi.py \
  tests/test_network.py tests/test_bigint.py tests/test_struct_layouts.py -q -m "not gpu" 2>&1 | grep -E "runtime error|AddressSanitizer|passed|failed|SUMMARY"
