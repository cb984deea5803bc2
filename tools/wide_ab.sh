#!/bin/bash
# A/B of the wide kernel against the halo kernel on the stride-1 3x3 layers (GPU box): production rule, then CELLSEG_WIDE = 1 (off) /
# 4 / 6 / 8 (forced pixel-tile height) through the A/B flavour of the library.
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
SH=${SH:-"l2_3x3 l3_3x3 l4_3x3 dec_3x3_2048_1024 dec_3x3_1024_512 dec_3x3_512_256 dec_3x3_256_128"}
OUT=${OUT:-gpurun_out/wide_mb.log}
echo "== prod rule" > $OUT
VARIANT=1 ONLY=none ITERS=30 timeout -k 10 200 python tools/conv_microbench.py $SH >> $OUT 2>&1
for w in ${KNOBS:-1 4 6 8}; do
  echo "== CELLSEG_WIDE=$w" >> $OUT
  CELLSEG_LIB_FLAVOUR=ab CELLSEG_WIDE=$w VARIANT=1 ONLY=none ITERS=30 timeout -k 10 200 python tools/conv_microbench.py $SH >> $OUT 2>&1
done
grep -v "amdgpu.ids" $OUT
