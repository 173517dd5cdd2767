#!/bin/bash
# Round-2 GPU call A: kernel A/B of the F(4,3) tiles, parity on the BASELINE configs, bench line, serialised kernel trace.
R=$PWD
mkdir -p $R/gpurun_out
cd $R
for sh in 0 1 2 3 5 6 7 8; do timeout 120 tools/bin/conv_bench 5 $sh wino43; done > gpurun_out/r02_conv_bench_w43.log 2>&1
echo "conv_bench rc=$?"
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s > gpurun_out/r02_gpu_configs.log 2>&1
echo "configs rc=$?"; tail -3 gpurun_out/r02_gpu_configs.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_gpu_parity.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/r02_gpu_parity.log
timeout 900 python bench.py --profile-out gpurun_out/r02_per_op_profile.json > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_bench_1gpu.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_rocprof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_rocprof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split --lanes 0 > $R/gpurun_out/r02_rocprof.log 2>&1
echo "kernel-trace rc=$?"
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/r02_rocprof/*/*results.db gpurun_out/r02_rocprof/*results.db 2>/dev/null | head -1) --forwards 5 > gpurun_out/r02_kernel_stats.md 2> gpurun_out/r02_kernel_stats.err
echo "summary rc=$?"
rm -rf gpurun_out/r02_rocprof/*/*.db gpurun_out/r02_rocprof/*.db
