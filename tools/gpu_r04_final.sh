#!/bin/bash
# Round-4 evidence run on the GPU box: full GPU test-suite, smoke, bench lines (1080p, 4K T=6, other configs, files-to-files),
# serialised kernel trace, PMC passes.
R=$PWD
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s > $O/r04_gpu_tests_full.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/r04_gpu_tests_full.log | tail -3
timeout 300 python __graft_entry__.py smoke > $O/r04_smoke.log 2>&1; echo "smoke rc=$?"; grep -i smoke $O/r04_smoke.log | tail -2
timeout 900 python bench.py --profile-out $O/r04_per_op_profile.json > $O/r04_bench_1gpu.json 2> $O/r04_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 $O/r04_bench_1gpu.json
timeout 900 python bench.py --no-cpu-baseline --workload 4k_4x4_T6 > $O/r04_bench_4k_t6.json 2>> $O/r04_bench_1gpu.err; cut -c1-200 $O/r04_bench_4k_t6.json
for wl in 256 vimeo_b8 photos; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 > $O/r04_bench_$wl.json 2>> $O/r04_bench_1gpu.err; cut -c1-160 $O/r04_bench_$wl.json; done
timeout 600 python bench.py --workload cli_1080p_T3 > $O/r04_bench_cli_1080p_t3.json 2>> $O/r04_bench_1gpu.err; cut -c1-200 $O/r04_bench_cli_1080p_t3.json
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 1080p_2x2_T3 --scaling strong --steps 3 > $O/r04_bench_1080p_t3_strong_1gpu.json 2>> $O/r04_bench_1gpu.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $O/r04_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls $O/rocprof/*/*results.db $O/rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > $O/r04_kernel_stats.md 2> $O/r04_kernel_stats.err
echo "summary rc=$?"; head -14 $O/r04_kernel_stats.md
rm -rf $O/rocprof
BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $O/pmc > $O/r04_pmc.log 2>&1
python tools/pmc_summary.py $O/pmc --md $O/r04_pmc_summary.md --json $O/r04_pmc_conv.json
echo "pmc rc=$?"; grep -E "^## |MFMA pipe|FETCH_SIZE|effective clock" $O/r04_pmc_summary.md | head -40
rm -rf $O/pmc/*/*.db
du -sh $O
