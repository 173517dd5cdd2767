#!/bin/bash
# Round-end evidence run on the GPU box (ONE tree, one gpurun call):  tools/gpu_evidence.sh <tag>   -> gpurun_out/<tag>/<tag>_*
#   1. the bench line with parity, per-op profile and CPU baseline        (<tag>_bench_1gpu.json, <tag>_per_op_profile.json)
#   2. rocprofv3 --kernel-trace --stats of a serialised forward            (<tag>_kernel_stats.md)
#   3. the PMC passes, one rocprofv3 process each (tools/gpu_pmc.sh)        (<tag>_pmc_summary.md, <tag>_pmc_conv.json)
#   4. the other BASELINE configs as bench lines, files-to-files, strong mode on one rank
#   5. the driver's sequence: smoke() and the full `pytest -m gpu`          (<tag>_smoke.log, <tag>_gpu_tests.log)
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ (tracked); steps can be skipped with SKIP="pmc tests ...".
R=$PWD; TAG=${1:-rXX}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
skip() { [[ " $SKIP " == *" $1 "* ]]; }
t0=$(date +%s); lap() { t1=$(date +%s); echo "[$1: $((t1-t0)) s]"; t0=$t1; }
timeout 900 python bench.py --profile-out $O/${TAG}_per_op_profile.json > $O/${TAG}_bench_1gpu.json 2> $O/${TAG}_bench.err
echo "bench rc=$?"; cut -c1-220 $O/${TAG}_bench_1gpu.json; lap bench
if ! skip trace; then
  cd /tmp && export TMPDIR=/tmp; rm -rf $O/rocprof
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $O/${TAG}_rocprof.log 2>&1
  echo "kernel-trace rc=$?"; cd $R
  python tools/rocprof_summary.py $(ls $O/rocprof/*/*results.db $O/rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > $O/${TAG}_kernel_stats.md 2> $O/${TAG}_kernel_stats.err
  echo "summary rc=$?"; head -14 $O/${TAG}_kernel_stats.md; rm -rf $O/rocprof; lap trace
fi
if ! skip pmc; then
  BENCH_ARGS="--lanes 0" tools/gpu_pmc.sh $O/pmc > $O/${TAG}_pmc.log 2>&1
  python tools/pmc_summary.py $O/pmc --md $O/${TAG}_pmc_summary.md --json $O/${TAG}_pmc_conv.json
  echo "pmc rc=$?"; grep -E "^## |MFMA pipe|FETCH_SIZE|effective clock" $O/${TAG}_pmc_summary.md | head -40; rm -rf $O/pmc/*/*.db; lap pmc
fi
if ! skip configs; then
  timeout 400 python bench.py --no-cpu-baseline --no-split --workload 4k_4x4_T6 > $O/${TAG}_bench_4k_t6.json 2>> $O/${TAG}_bench.err; cut -c1-200 $O/${TAG}_bench_4k_t6.json
  for wl in 256 vimeo_b8 photos; do timeout 200 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 --profile-out $O/${TAG}_per_op_profile_$wl.json > $O/${TAG}_bench_$wl.json 2>> $O/${TAG}_bench.err; cut -c1-160 $O/${TAG}_bench_$wl.json; done
  timeout 300 python bench.py --workload cli_1080p_T3 > $O/${TAG}_bench_cli_1080p_t3.json 2>> $O/${TAG}_bench.err; cut -c1-200 $O/${TAG}_bench_cli_1080p_t3.json
  timeout 200 python bench.py --no-cpu-baseline --no-split --workload 1080p_2x2_T3 --scaling strong --steps 3 > $O/${TAG}_bench_1080p_t3_strong_1gpu.json 2>> $O/${TAG}_bench.err; lap configs
fi
if ! skip tests; then
  timeout 60 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; grep -i smoke $O/${TAG}_smoke.log | tail -2
  timeout 900 python -m pytest tests -m gpu -q -s -rs > $O/${TAG}_gpu_tests_full.log 2>&1
  echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/${TAG}_gpu_tests_full.log > $O/${TAG}_gpu_tests.log; rm -f $O/${TAG}_gpu_tests_full.log; grep -i "passed\|failed" $O/${TAG}_gpu_tests.log | tail -3; lap tests
fi
du -sh $O
