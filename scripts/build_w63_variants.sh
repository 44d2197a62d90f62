#!/bin/bash
# Timing-only build variants of tonal_wino63.hip (ABL_MACRO=V6_ABL for the NT kernel, T6_ABL for the weight-gradient kernel; V_EXTRA
# for further -D switches): one libtonal_hip.so per variant under build/variants/, selected at run time with TONAL_HIP_LIB.
# Usage: [TAG=x] scripts/build_w63_variants.sh 0 1 2 4
set -e
cd "$(dirname "$0")/../decode_tonal_langauge_amd/csrc"
mkdir -p ../../build/variants
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops -D${ABL_MACRO:-V6_ABL}=$v ${V_EXTRA} -c tonal_wino63.hip -o ../../build/variants/w63_${TAG}$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib63_${TAG}$v.so tonal_gemm.o tonal_misc.o tonal_signal.o tonal_lite.o tonal_steps.o tonal_wino.o tonal_wino43_tn.o tonal_wino43v.o ../../build/variants/w63_${TAG}$v.o
done
