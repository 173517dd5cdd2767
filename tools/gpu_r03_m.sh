#!/bin/bash
O=gpurun_out/r03m
mkdir -p $O
for wl in 256 photos vimeo_b8; do for L in 1 0 1 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --wino2d $L --steps 40 --profile-out $O/per_op_${wl}_w2d$L.json > $O/bench_${wl}_w2d$L.json 2> $O/bench.err
  echo "$wl wino2d $L: $(python -c "import json;d=json.load(open('$O/bench_${wl}_w2d$L.json'));print(d['ms_per_step'], d['kernel_ms_per_step'])")"
done; done
python tools/prof_compare.py $O/per_op_256_w2d1.json $O/per_op_256_w2d0.json 200 | grep -E "tile 8[0-9]{3} |conv total"
