SH="l1_1x1_64_256 l2_1x1_128_512 l2_1x1_512_128 l3_1x1_256_1024 l3_1x1_1024_256 l4_1x1_512_2048 l4_1x1_2048_512 l3_1x1_s2_512_1024 l2_3x3_s2 l4_1x1_1024_512"
for spec in 0 1; do
  echo "== CELLSEG_WGRAD_SPEC=$spec"
  CELLSEG_LIB_FLAVOUR=ab CELLSEG_WGRAD_SPEC=$spec ONLY=wgrad_b BATCH=${BATCH:-4} ITERS=20 python tools/conv_microbench.py $SH 2>&1 | grep -v amdgpu.ids | sed 's/fwd_pk.*//'
done
