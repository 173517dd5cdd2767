#!/bin/bash
R=$PWD
cd $R
for sh in 0 2; do timeout 120 tools/bin/conv_bench 5 $sh "wino43 q16"; done > gpurun_out/r02f_conv_bench_ablation.log 2>&1
grep -E "==|f4  |f260|f4100|f8196|f16388|f3844" gpurun_out/r02f_conv_bench_ablation.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "graph_replay or published_64 or aux" > gpurun_out/r02f_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r02f_gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-split --profile-out gpurun_out/r02f_per_op_profile.json > gpurun_out/r02f_bench_1gpu.json 2> gpurun_out/r02f_bench_1gpu.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r02f_bench_1gpu.json
