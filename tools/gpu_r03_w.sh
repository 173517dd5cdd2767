#!/bin/bash
# round 3, call W: warp_vec_kernel with the eight rows' flows requested up front
O=gpurun_out/r03w
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r3.py tests/test_gpu_configs.py -m gpu -x -q -k "warp or tile_960x576 or deterministic or large_flows or stages or 1080p_2x2 or graph_replay" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | tail -2
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_$i.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline_warp']['frac'])"; done
