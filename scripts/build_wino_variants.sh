#!/bin/bash
# Build variants of libtonal_hip.so from an edited copy of tonal_wino.hip into build/var/lib_<name>.so
# (perf experiments; select one with TONAL_HIP_LIB=build/var/lib_<name>.so).
#   scripts/build_wino_variants.sh 'name|-DFLAGS|sed-script' ...
set -e
cd "$(dirname "$0")/.."
SRC=decode_tonal_langauge_amd/csrc
OUT=build/var
mkdir -p $OUT
make -s -C $SRC >/dev/null
for spec in "$@"; do
  IFS='|' read -r name flags script <<< "$spec"
  d=$(mktemp -d)
  cp $SRC/tonal_wino.hip $SRC/tonal_common.h $d/
  sed -i "s#\"../../include/tonal_hip.h\"#\"$PWD/include/tonal_hip.h\"#" $d/tonal_common.h
  [ -n "$script" ] && sed -i "$script" $d/tonal_wino.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $d/tonal_wino.hip -o $d/wino.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $d/wino.o \
      $SRC/tonal_gemm.o $SRC/tonal_misc.o $SRC/tonal_signal.o $SRC/tonal_lite.o $SRC/tonal_steps.o
  rm -rf $d
done
ls $OUT
