#!/bin/bash
# round 3, call U: engine with the raw-staging nested-Winograd tiles
O=gpurun_out/r03u
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py tests/test_gpu_parity.py -m gpu -x -q -s -k "wino2d or nested or tile_960x576 or deterministic or f43_tile" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | grep "exercised\|passed\|failed\|Error\|assert" | tail -12
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_$i.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"; done
