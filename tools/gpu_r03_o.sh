#!/bin/bash
O=gpurun_out/r03o
mkdir -p $O
for S in 2 1 5 13; do timeout 200 tools/bin/conv_bench 7 $S "wino43 q16 nh1 4x64x64 t1x1 f32772,wino43 q16 nh1 4x64x64 t1x1 f1081348,wino43 q8 nh1 8x32x64 t1x1 f32772,wino43 q8 nh1 8x32x64 t1x1 f1081348,wino43 q8 8x32x32 t1x1 f32772,wino43 q8 8x32x32 t1x1 f1081348,wino43 q16 4x64x64 t2x1 f32772,wino43 q16 4x64x64 t2x1 f1081348"; done 2>&1 | grep -E "==|TF/s" | tee $O/conv_bench_pin.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_$i.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'], d['value'], d['roofline']['class_ms_per_step'])"; done
