#!/bin/bash
# time the forward V kernel of every build variant given (conv2 shape, batch 256): scripts/run_v_variants.sh 0 1 2 ...
for v in "$@"; do
  echo "== V4_ABL=$v"
  TONAL_HIP_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python scripts/bench_conv.py --stages ${STAGES:-2} --passes ${PASSES:-fwd} --iters 3 || exit 1
done
