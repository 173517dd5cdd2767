#!/bin/bash
# LDS counters for tools/bin/conv_bench variants.  Usage: tools/gpu_pmc_conv2.sh <outdir> <shape> <filter>
R=$PWD
OUT=${1:-gpurun_out/pmc_conv2}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o pmc -- $R/tools/bin/conv_bench 2 ${2:-0} "${3:-128x128}" > $OUT/sq2.log 2>&1
echo "sq2 rc=$?"
