#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_r3.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "graph_replay or tile_960x576 or 1080p_2x2 or deterministic" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_$i.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"; done
