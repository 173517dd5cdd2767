#!/bin/bash
# round 5, call G: bench line + kernel trace + PMC passes of the final source id (a comment in conv_wino2d_impl.h changed after tools/gpu_r05_final.sh)
R=$PWD; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
timeout 600 python bench.py --profile-out $O/r05_per_op_profile.json > $O/r05_bench_1gpu.json 2> $O/r05_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 $O/r05_bench_1gpu.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lanes 0 > $O/r05_rocprof.log 2>&1
cd $R
python tools/rocprof_summary.py $(ls $O/rocprof/*/*results.db $O/rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > $O/r05_kernel_stats.md 2> $O/r05_kernel_stats.err
head -8 $O/r05_kernel_stats.md; rm -rf $O/rocprof
BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $O/pmc > $O/r05_pmc.log 2>&1
python tools/pmc_summary.py $O/pmc --md $O/r05_pmc_summary.md --json $O/r05_pmc_conv.json
echo "pmc rc=$?"; grep -E "MFMA pipe|effective clock" $O/r05_pmc_summary.md | head -4
rm -rf $O/pmc/*/*.db
timeout 60 python __graft_entry__.py smoke 2>&1 | grep -i "smoke:"
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "1080p or tile_960 or vimeo or 256" 2>&1 | grep -i "passed\|failed" | tail -2
