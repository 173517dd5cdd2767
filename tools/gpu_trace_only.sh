#!/bin/bash
# serialised kernel trace of the bench forward + its summary (the middle part of tools/gpu_r02_final.sh)
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $R/gpurun_out/r02_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/r02_rocprof/*/*results.db gpurun_out/r02_rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > gpurun_out/r02_kernel_stats.md 2> gpurun_out/r02_kernel_stats.err
echo "summary rc=$?"; head -8 gpurun_out/r02_kernel_stats.md; cat gpurun_out/r02_kernel_stats.err
rm -rf gpurun_out/r02_rocprof
