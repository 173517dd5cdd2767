#!/bin/bash
# One SQ PMC pass over a bench forward: per-dispatch clock / MFMA busy (tools/pmc_table.py reads it).
R=$PWD
OUT=${1:-gpurun_out/pmc_fwd}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq1.log 2>&1
echo "sq1 rc=$?"
