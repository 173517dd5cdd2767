#!/bin/bash
# round 3, call W: conv_wino2d_kernel eligibility no longer tied to the 1-D kernel's thresholds (small frames)
O=gpurun_out/r03w
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | tail -2
for wl in 256 vimeo_b8 photos; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 > $O/bench_$wl.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', d['ms_per_step'], d['kernel_ms_per_step'])"; done
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_1.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline_warp']['frac'])"
