#!/bin/bash
# Round-2 GPU call B: stagger / NH=1 variants of the F(4,3) kernel, fused small ops + new warp kernel (parity + time)
R=$PWD
mkdir -p $R/gpurun_out
cd $R
for sh in 2 1 5 0 3 7 8; do timeout 120 tools/bin/conv_bench 5 $sh "wino43 q16"; done > gpurun_out/r02b_conv_bench_w43.log 2>&1
echo "conv_bench rc=$?"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02b_gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r02b_gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-split --profile-out gpurun_out/r02b_per_op_profile.json > gpurun_out/r02b_bench_1gpu.json 2> gpurun_out/r02b_bench_1gpu.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r02b_bench_1gpu.json
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 > gpurun_out/r02b_bench_256.json 2>> gpurun_out/r02b_bench_1gpu.err
cut -c1-200 gpurun_out/r02b_bench_256.json
