#!/bin/bash
# Timing-only build variants of tonal_wino43v.hip (V4_ABL bit switches; ABL_MACRO=T4V_ABL for the weight-gradient kernel): one libtonal_hip.so per variant under
# build/variants/, selected at run time with TONAL_HIP_LIB.  Usage: scripts/build_v_variants.sh 0 1 2 4 8 16
set -e
cd "$(dirname "$0")/../decode_tonal_langauge_amd/csrc"
mkdir -p ../../build/variants
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -D${ABL_MACRO:-V4_ABL}=$v ${V_EXTRA} -c tonal_wino43v.hip -o ../../build/variants/w43v_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib_$v.so tonal_gemm.o tonal_misc.o tonal_signal.o tonal_lite.o tonal_steps.o tonal_wino.o tonal_wino43_tn.o ../../build/variants/w43v_$v.o
done
