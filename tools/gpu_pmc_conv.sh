#!/bin/bash
# PMC pass over tools/bin/conv_bench (one shape, variants filtered by name).  Usage: tools/gpu_pmc_conv.sh <outdir> <shape> <filter>
R=$PWD
OUT=${1:-gpurun_out/pmc_conv}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o pmc -- $R/tools/bin/conv_bench 2 ${2:-0} "${3:-128x128}" > $OUT/sq1.log 2>&1
echo "sq1 rc=$?"
