#!/bin/bash
# PMC passes over one bench forward (run on the GPU box via gpurun).  Each pass is its own process:
# counters are never combined with sys/runtime traces (pool rule) and FETCH_SIZE / WRITE_SIZE need
# separate passes (TCC slots).  Usage: [BENCH_ARGS="--precision 2"] tools/gpu_pmc.sh <outdir>
R=$PWD
OUT=${1:-$R/gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
BENCH="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split $BENCH_ARGS"
pass() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
ls -R $OUT | head -40
