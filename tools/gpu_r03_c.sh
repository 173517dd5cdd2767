#!/bin/bash
# round 3, call C: Q8 tiles in the engine - parity (every tile shape, T = 6 tree) + the bench line + per-op profile
O=gpurun_out/r03c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py -m gpu -x -q -s -k "tile_shape or depth6" > $O/gpu_r3_tests.log 2>&1
echo "r3 tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_r3_tests.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-split --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench_1gpu.err
echo "bench rc=$?"; cut -c1-300 $O/bench_1gpu.json
python tools/prof_compare.py $O/per_op_profile.json profiles/r02_per_op_profile.json 45 2>&1 | tail -48
