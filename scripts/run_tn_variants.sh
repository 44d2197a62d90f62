#!/bin/bash
# time the tilings of the V-form weight-gradient kernel for every build variant given (conv2 / conv3 shape, batch 256):
# ABL_MACRO=T8_ABL scripts/build_v_variants.sh 1 2 ... && BMS=127,128 scripts/run_tn_variants.sh 1 2 ...
for v in "$@"; do
  echo "== variant $v"
  TONAL_HIP_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python scripts/check_tn_bm.py --batch 256 --channels 128 --iters 3 --stages ${STAGES:-2} --bms ${BMS:-64,128} | grep "^bm" || exit 1
done
