#!/bin/bash
# Round-3 evidence run on the GPU box: full GPU test-suite, smoke, bench lines (1080p, 4K T=6, other configs), warp experiment with
# zero flows, serialised kernel trace, PMC passes.
R=$PWD
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s > $O/r03_gpu_tests_full.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/r03_gpu_tests_full.log | tail -3
timeout 300 python __graft_entry__.py smoke > $O/r03_smoke.log 2>&1; echo "smoke rc=$?"; grep smoke $O/r03_smoke.log
timeout 900 python bench.py --profile-out $O/r03_per_op_profile.json > $O/r03_bench_1gpu.json 2> $O/r03_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 $O/r03_bench_1gpu.json
timeout 900 python bench.py --no-cpu-baseline --workload 4k_4x4_T6 > $O/r03_bench_4k_t6.json 2>> $O/r03_bench_1gpu.err; cut -c1-200 $O/r03_bench_4k_t6.json
for wl in 256 vimeo_b8 photos; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 > $O/r03_bench_$wl.json 2>> $O/r03_bench_1gpu.err; cut -c1-160 $O/r03_bench_$wl.json; done
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 1080p_2x2_T3 --scaling strong --steps 3 > $O/r03_bench_1080p_t3_strong_1gpu.json 2>> $O/r03_bench_1gpu.err
for FS in 1.0 0.0; do timeout 300 python bench.py --no-cpu-baseline --no-split --flow-scale $FS --steps 10 > $O/r03_bench_flowscale_$FS.json 2>> $O/r03_bench_1gpu.err; python -c "import json;d=json.load(open('$O/r03_bench_flowscale_$FS.json'));print('flow-scale $FS:', d['ms_per_step'], d['roofline_warp'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $O/r03_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls $O/rocprof/*/*results.db $O/rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > $O/r03_kernel_stats.md 2> $O/r03_kernel_stats.err
echo "summary rc=$?"; head -12 $O/r03_kernel_stats.md
rm -rf $O/rocprof
BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $O/pmc > $O/r03_pmc.log 2>&1
python tools/pmc_summary.py $O/pmc --md $O/r03_pmc_summary.md --json $O/r03_pmc_conv.json
echo "pmc rc=$?"; grep -E "^## |MFMA pipe|FETCH_SIZE|effective clock" $O/r03_pmc_summary.md | head -40
BENCH_ARGS="--lanes 0 --flow-scale 0.0" tools/gpu_pmc.sh $O/pmc_zero > $O/r03_pmc_zero.log 2>&1
python tools/pmc_summary.py $O/pmc_zero --md $O/r03_pmc_summary_zero_flows.md --json $O/r03_pmc_conv_zero.json
rm -rf $O/pmc/*/*.db $O/pmc_zero/*/*.db
du -sh $O
