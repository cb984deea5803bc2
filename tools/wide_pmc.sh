#!/bin/bash
# PMC passes on the wide / halo kernels of one microbench shape (GPU box).  Usage: SH="l3_3x3" W=6 bash tools/wide_pmc.sh
set -o pipefail
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$PWD
SH=${SH:-l3_3x3}
W=${W:-6}
OUT=$ROOTD/gpurun_out/pmc_wide_${SH}_$W
rm -rf $OUT; mkdir -p $OUT
cd $ROOTD
export CELLSEG_LIB_FLAVOUR=ab CELLSEG_WIDE=$W ONLY=none ITERS=5
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p1 -- python tools/conv_microbench.py $SH > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $OUT/p2 -- python tools/conv_microbench.py $SH > $OUT/p2.log 2>&1
python tools/pmc_summary.py $OUT conv2_ > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
