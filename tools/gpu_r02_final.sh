#!/bin/bash
# Round-2 evidence run on the GPU box: full GPU test-suite, smoke, bench lines, serialised kernel trace, PMC passes.
R=$PWD
mkdir -p $R/gpurun_out
cd $R
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r02_gpu_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?"; grep smoke gpurun_out/r02_smoke.log
timeout 900 python bench.py --profile-out gpurun_out/r02_per_op_profile.json > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_bench_1gpu.json
for wl in 256 vimeo_b8 photos; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 > gpurun_out/r02_bench_$wl.json 2>> gpurun_out/r02_bench_1gpu.err; cut -c1-160 gpurun_out/r02_bench_$wl.json; done
python bench.py --gpus 2 > gpurun_out/r02_bench_2gpu_refusal.log 2>&1; echo "--gpus 2 on a 1-GPU box: rc=$?" >> gpurun_out/r02_bench_2gpu_refusal.log; tail -2 gpurun_out/r02_bench_2gpu_refusal.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $R/gpurun_out/r02_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/r02_rocprof/*/*results.db gpurun_out/r02_rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > gpurun_out/r02_kernel_stats.md 2> gpurun_out/r02_kernel_stats.err
echo "summary rc=$?"; head -8 gpurun_out/r02_kernel_stats.md
rm -rf gpurun_out/r02_rocprof
BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $R/gpurun_out/r02_pmc > gpurun_out/r02_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r02_pmc --md gpurun_out/r02_pmc_summary.md --json gpurun_out/r02_pmc_conv.json
echo "pmc rc=$?"; grep -E "^## |MFMA pipe|FETCH_SIZE|effective clock" gpurun_out/r02_pmc_summary.md | head -40
rm -rf gpurun_out/r02_pmc/*/*.db
du -sh gpurun_out/r02_pmc
