#!/bin/bash
# Build variants of libtonal_hip.so with ONE source file edited / re-flagged, into build/var/lib_<name>.so
# (perf experiments only; select one with TONAL_HIP_LIB=build/var/lib_<name>.so).
#   scripts/build_variant.sh tonal_wino43_tn.hip 'name|-DFLAGS|sed-script' ...
set -e
cd "$(dirname "$0")/.."
SRC=decode_tonal_langauge_amd/csrc
OUT=build/var
FILE=$1; shift
mkdir -p $OUT
make -s -C $SRC >/dev/null
others=$(ls $SRC/*.o | grep -v "/${FILE%.hip}.o")
for spec in "$@"; do
  IFS='|' read -r name flags script <<< "$spec"
  d=$(mktemp -d)
  cp $SRC/$FILE $SRC/tonal_common.h $d/
  sed -i "s#\"../../include/tonal_hip.h\"#\"$PWD/include/tonal_hip.h\"#" $d/tonal_common.h
  [ -n "$script" ] && sed -i "$script" $d/$FILE
  extra=""
  [ "$FILE" = "tonal_signal.hip" ] && extra="-ffp-contract=off"
  [ "$FILE" = "tonal_wino43_tn.hip" ] && extra="-Xclang -target-feature -Xclang -packed-fp32-ops"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra $flags -c $d/$FILE -o $d/v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $d/v.o $others
  rm -rf $d
done
ls $OUT
