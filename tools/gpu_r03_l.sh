#!/bin/bash
R=$PWD
O=$R/gpurun_out/r03l
mkdir -p $O
timeout 200 tools/bin/conv_bench 5 0 "wino2d q8 8x32 f68,wino2d q8 8x64 f68,wino2d q16 4x32 f68,wino2d q8 8x32 f1028,wino43 q8 nh1 8x32x64 t1x1 f32772" 2>&1 | grep -E "==|TF/s"
timeout 200 tools/bin/conv_bench 5 7 "wino2d q8 8x32 f68,wino2d q8 8x64 f68,wino2d q16 4x32 f68,wino43 q8 8x32x32 t1x1 f32772" 2>&1 | grep -E "==|TF/s"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -o pmc -- $R/tools/bin/conv_bench 1 0 "wino2d q8 8x32 f68,wino2d q16 4x32 f68" > $O/sq2.log 2>&1; echo "sq2 rc=$?"
cd $R; python tools/pmc_raw_table.py $O/sq2 conv_ | tail -4
