#!/bin/bash
# Build variants of one kernel file (macro switches on the command line), one libtonal_hip.so per variant under
# build/variants/, selected at run time with TONAL_HIP_LIB:
#   scripts/build_w_variants.sh tonal_wino name1 "-DW4_SCHED=3" name2 "-DW4_SCHED=5" ...
set -e
cd "$(dirname "$0")/../decode_tonal_langauge_amd/csrc"
mkdir -p ../../build/variants
src=$1; shift
objs=""
for o in tonal_gemm tonal_misc tonal_signal tonal_lite tonal_steps tonal_wino tonal_wino43_tn tonal_wino43v; do
  [ "$o" != "$src" ] && objs="$objs $o.o"
done
extra=""
[ "$src" = "tonal_wino43_tn" ] && extra="-Xclang -target-feature -Xclang -packed-fp32-ops"
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $extra $f -c $src.hip -o ../../build/variants/${src}_$n.o 2>&1 | grep -v "recognized feature" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/lib_$n.so $objs ../../build/variants/${src}_$n.o
done
