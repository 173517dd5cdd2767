#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/quick_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/quick_gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-split --profile-out gpurun_out/quick_per_op_profile.json > gpurun_out/quick_bench_1gpu.json 2> gpurun_out/quick_bench_1gpu.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/quick_bench_1gpu.json
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 > gpurun_out/quick_bench_256.json 2>> gpurun_out/quick_bench_1gpu.err
cut -c1-200 gpurun_out/quick_bench_256.json
