#!/bin/bash
O=gpurun_out/r03v
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 40 --profile-out $O/per_op_256.json > $O/bench_256.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench_256.json'));print(d['ms_per_step'], d['kernel_ms_per_step'])"
