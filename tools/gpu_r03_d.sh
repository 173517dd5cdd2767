#!/bin/bash
# round 3, call D: coarse decoder levels on the side lane (lanes 2) vs round-2 order (lanes 1) + graph-replay race tests
O=gpurun_out/r03d
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py tests/test_gpu_parity.py -m gpu -x -q -s -k "tile_shape or graph_replay or deterministic or fused_rgb" > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_tests.log | tail -6
for L in 2 1 2 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-split --lanes $L --steps 20 > $O/bench_lanes$L.json 2> $O/bench.err
  echo "lanes $L: $(python -c "import json;d=json.load(open('$O/bench_lanes$L.json'));print(d['ms_per_step'], d['value'], d['roofline']['class_ms_per_step'], d['kernel_ms_per_step'])")"
done
