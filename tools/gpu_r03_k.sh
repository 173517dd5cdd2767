#!/bin/bash
# PMC passes of conv_bench on the 528 -> 128 layer: nested Winograd vs the 1-D F(4,3) tile (where do the cycles go?)
R=$PWD
O=$R/gpurun_out/r03k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
V="wino2d q8 8x32 f68,wino2d q8 8x32 f1028,wino43 q8 nh1 8x32x64 t1x1 f32772,wino2d q8 8x64 f68"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o pmc -- $R/tools/bin/conv_bench 2 0 "$V" > $O/$name.log 2>&1; echo "$name rc=$?"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
pass ta TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
cd $R
for p in sq1 sq2 ta tcp; do echo "== $p"; python tools/pmc_raw_table.py $O/$p conv_ 2>&1 | tail -8; done
python tools/pmc_table.py $O/sq1 conv_ | tail -8
rm -rf $O/*/*.db
