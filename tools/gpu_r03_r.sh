#!/bin/bash
O=gpurun_out/r03r
mkdir -p $O
for F in 31 30 31 30 27; do
  timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --fuse $F > $O/bench_f$F.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_f$F.json'));print('fuse $F:', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
done
