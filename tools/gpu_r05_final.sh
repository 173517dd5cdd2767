#!/bin/bash
# Round-5 evidence run on the GPU box (one tree, one call): bench lines (1080p with parity + CPU baseline, 4K T=6, other configs,
# files-to-files, strong 1-GPU), serialised kernel trace, PMC passes, then the full GPU suite on the default library, the opt-in-family
# tests on the FILM_EXTRA_FAMILIES flavour and smoke.
R=$PWD
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
t0=$(date +%s); lap() { t1=$(date +%s); echo "[$1: $((t1-t0)) s]"; t0=$t1; }
# same-box A/B of the split-K index math in front of the first DMA request (tools/bin/w2d_bench_before = the kernel of commit e9480fd)
if [ -x tools/bin/w2d_bench_before ]; then
  for sh in 15 10 18 7 16 6 0 1; do for which in w2d_bench_before w2d_bench w2d_bench_before w2d_bench; do echo "## $which"; timeout 60 tools/bin/$which 10 $sh "=w2d 64,=w2d 32"; done; done > $O/r05_w2d_prologue_fix.log 2>&1
  grep -E "^##|^==|w2d (64|32) " $O/r05_w2d_prologue_fix.log | cut -c1-120 | head -100; lap w2d_ab
fi
timeout 600 python bench.py --profile-out $O/r05_per_op_profile.json > $O/r05_bench_1gpu.json 2> $O/r05_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 $O/r05_bench_1gpu.json; lap bench
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rocprof
timeout 400 rocprofv3 --kernel-trace --stats -d $O/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lanes 0 > $O/r05_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls $O/rocprof/*/*results.db $O/rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > $O/r05_kernel_stats.md 2> $O/r05_kernel_stats.err
echo "summary rc=$?"; head -14 $O/r05_kernel_stats.md
rm -rf $O/rocprof; lap trace
BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $O/pmc > $O/r05_pmc.log 2>&1
python tools/pmc_summary.py $O/pmc --md $O/r05_pmc_summary.md --json $O/r05_pmc_conv.json
echo "pmc rc=$?"; grep -E "^## |MFMA pipe|FETCH_SIZE|effective clock" $O/r05_pmc_summary.md | head -40
rm -rf $O/pmc/*/*.db; lap pmc
timeout 400 python bench.py --no-cpu-baseline --workload 4k_4x4_T6 > $O/r05_bench_4k_t6.json 2>> $O/r05_bench_1gpu.err; cut -c1-200 $O/r05_bench_4k_t6.json
for wl in 256 vimeo_b8 photos; do timeout 200 python bench.py --no-cpu-baseline --workload $wl --steps 30 --profile-out $O/r05_per_op_profile_$wl.json > $O/r05_bench_$wl.json 2>> $O/r05_bench_1gpu.err; cut -c1-160 $O/r05_bench_$wl.json; done
timeout 300 python bench.py --workload cli_1080p_T3 > $O/r05_bench_cli_1080p_t3.json 2>> $O/r05_bench_1gpu.err; cut -c1-200 $O/r05_bench_cli_1080p_t3.json
timeout 200 python bench.py --no-cpu-baseline --workload 1080p_2x2_T3 --scaling strong --steps 3 > $O/r05_bench_1080p_t3_strong_1gpu.json 2>> $O/r05_bench_1gpu.err; lap other_benches
SEGV_BT_OUT=$O/segv.txt LD_PRELOAD=$R/tools/bin/segv_bt.so timeout 600 python -m pytest tests -m gpu -q -s -p no:faulthandler > $O/r05_gpu_tests_full.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/r05_gpu_tests_full.log | grep -i "passed\|failed" | tail -3; head -5 $O/segv.txt 2>/dev/null; lap tests
timeout 60 python __graft_entry__.py smoke > $O/r05_smoke.log 2>&1; echo "smoke rc=$?"; grep -i smoke $O/r05_smoke.log | tail -2
FILM_EXTRA_FAMILIES=1 timeout 420 python -m pytest tests -m gpu -q -s -k "precision or halo_kernels or winograd_kernel_on or second_weight_set" > $O/r05_gpu_tests_extra_full.log 2>&1
echo "extra tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/r05_gpu_tests_extra_full.log | grep -i "passed\|failed" | tail -3; lap extra_tests
du -sh $O
