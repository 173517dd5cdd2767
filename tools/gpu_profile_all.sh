#!/bin/bash
# Round-end evidence run on the GPU box: kernel trace + stats of bench.py, then the PMC passes.
# Usage: [BENCH_ARGS="--precision 2"] [SKIP_PMC=1] tools/gpu_profile_all.sh <tag>   (writes gpurun_out/<tag>_*)
R=$PWD
TAG=${1:-r01}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split $BENCH_ARGS > $R/gpurun_out/${TAG}_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/${TAG}_rocprof/*/*results.db gpurun_out/${TAG}_rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > gpurun_out/${TAG}_kernel_stats.md 2> gpurun_out/${TAG}_kernel_stats.err
echo "summary rc=$?"
[ -n "$SKIP_PMC" ] && exit 0
tools/gpu_pmc.sh $R/gpurun_out/${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmc --md gpurun_out/${TAG}_pmc_summary.md --json gpurun_out/${TAG}_pmc_conv.json
echo "pmc rc=$?"
