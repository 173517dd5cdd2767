#!/bin/bash
# PMC pass over tools/bin/w2d_bench (one shape, variants filtered by name).  Usage: tools/gpu_pmc_w2d.sh <outdir> <shape> <filter>
R=$PWD
OUT=${1:-gpurun_out/pmc_w2d}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o pmc -- $R/tools/bin/w2d_bench 3 ${2:-0} "${3:-w2f}" > $OUT/sq1.log 2>&1
echo "sq1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o pmc -- $R/tools/bin/w2d_bench 3 ${2:-0} "${3:-w2f}" > $OUT/sq2.log 2>&1
echo "sq2 rc=$?"
cd $R
python tools/pmc_table.py $OUT/sq1 conv_ | awk '!seen[$1]++ || 1' > $OUT/sq1_table.txt
python tools/pmc_raw_table.py $OUT/sq2 conv_ > $OUT/sq2_table.txt
