#!/bin/bash
# Build ablation variants of the GEMM file into gpurun_out/var/lib_<name>.so (perf experiments only).
set -e
cd "$(dirname "$0")/.."
SRC=decode_tonal_langauge_amd/csrc
OUT=build/var
mkdir -p $OUT
build() {  # name, sed-script
  d=$(mktemp -d)
  cp $SRC/*.hip $SRC/*.h $d/
  mkdir -p $d/../../include 2>/dev/null || true
  sed -i "s#\"../../include/tonal_hip.h\"#\"$PWD/include/tonal_hip.h\"#" $d/tonal_common.h
  sed -i "$2" $d/tonal_gemm.hip
  for f in tonal_gemm tonal_misc tonal_signal tonal_lite tonal_steps; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $d/$f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$1.so $d/*.o
  rm -rf $d
}
build base 's#XXXXNOPE##'
# no epilogue stores in the POOL path
build noepi 's#if (rowok \&\& colok) p.out\[prow \* (long long)p.ldo + col\] = o;#if (rowok \&\& colok \&\& o == 12345.678f) p.out[prow * (long long)p.ldo + col] = o;#'
# A tile loaded only for chunk 0 (no further A global loads / LDS stores)
build noA 's#if (j == 0 \&\& has_next_chunk) load_a(chunk + 1);#if (j == 0 \&\& has_next_chunk \&\& nsteps < 0) load_a(chunk + 1);#; s#if (j == J - 1 \&\& has_next_chunk) store_a(abuf ^ 1);#if (j == J - 1 \&\& has_next_chunk \&\& nsteps < 0) store_a(abuf ^ 1);#'
# no B loads / stores after the prologue
build noB 's#if (s + 2 < nsteps) load_b(rb_ld, c_ld, j_ld);#if (s + 2 < nsteps \&\& nsteps < 0) load_b(rb_ld, c_ld, j_ld);#; s#if (more) store_b(rb_st, bbuf ^ 1);#if (more \&\& nsteps < 0) store_b(rb_st, bbuf ^ 1);#'
ls -la $OUT
