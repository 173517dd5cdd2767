#!/bin/bash
# round 3, call W: warp_vec_kernel variants
O=gpurun_out/r03w
mkdir -p $O
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_$i.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline_warp']['frac'])"; done
