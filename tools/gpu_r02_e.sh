#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
cd $R
for sh in 0 2; do timeout 120 tools/bin/conv_bench 5 $sh "wino43 q16 4x64x64 t2x1"; done > gpurun_out/r02e_conv_bench_ablation.log 2>&1
cat gpurun_out/r02e_conv_bench_ablation.log
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02e_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r02e_gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-split --profile-out gpurun_out/r02e_per_op_profile.json > gpurun_out/r02e_bench_1gpu.json 2> gpurun_out/r02e_bench_1gpu.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r02e_bench_1gpu.json
