#!/bin/bash
# A/B of the wide kernel's 1x1 form against the ring kernel on the deep 1x1 layers (GPU box): CELLSEG_WIDE1 = 1 (off) / 4 / 6 / 8.
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
SH=${SH:-"l3_1x1_1024_256 l4_1x1_2048_512 l4_1x1_512_2048 l4_1x1_1024_512 l3_1x1_256_1024 l2_1x1_512_128"}
OUT=${OUT:-gpurun_out/wide1_mb.log}
echo "== prod rule" > $OUT
VARIANT=1 ONLY=none ITERS=30 timeout -k 10 200 python tools/conv_microbench.py $SH >> $OUT 2>&1
for w in ${KNOBS:-1 4 6}; do
  echo "== CELLSEG_WIDE1=$w" >> $OUT
  CELLSEG_LIB_FLAVOUR=ab CELLSEG_WIDE1=$w VARIANT=1 ONLY=none ITERS=30 timeout -k 10 200 python tools/conv_microbench.py $SH >> $OUT 2>&1
done
grep -v "amdgpu.ids" $OUT
