#!/bin/bash
# round 3, call A: new GPU tests + the 4K T=6 workload + the default bench line
R=$PWD
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_r3.py -m gpu -x -q -s > $O/gpu_r3_tests.log 2>&1
echo "r3 tests rc=$?"; tail -15 $O/gpu_r3_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-split --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench_1gpu.err
echo "bench rc=$?"; cut -c1-400 $O/bench_1gpu.json
timeout 900 python bench.py --no-cpu-baseline --workload 4k_4x4_T6 > $O/bench_4k_t6.json 2> $O/bench_4k_t6.err
echo "bench 4k rc=$?"; cut -c1-600 $O/bench_4k_t6.json; tail -3 $O/bench_4k_t6.err
timeout 300 python bench.py --no-cpu-baseline --workload 1080p_2x2_T3 --scaling strong --steps 3 > $O/bench_1080p_t3_strong.json 2> $O/bench_strong.err
echo "bench strong rc=$?"; cut -c1-400 $O/bench_1080p_t3_strong.json; tail -3 $O/bench_strong.err
