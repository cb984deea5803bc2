set -e
SH="l1_1x1_64_256 l2_1x1_256_128 l2_1x1_128_512 l2_1x1_512_128 l3_1x1_512_256 l3_1x1_256_1024 l3_1x1_1024_256 l4_1x1_1024_512 l4_1x1_512_2048 l4_1x1_2048_512 l3_1x1_s2_512_1024 l4_1x1_s2_1024_2048"
for tm in 2 3 4; do
  echo "== CELLSEG_RING_TM=$tm"
  CELLSEG_LIB_FLAVOUR=ab CELLSEG_RING_TM=$tm RESID=1 ONLY=none ITERS=30 python tools/conv_microbench.py $SH
done
echo "== auto"
RESID=1 ONLY=none ITERS=30 python tools/conv_microbench.py $SH
