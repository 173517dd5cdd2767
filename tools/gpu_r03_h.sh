#!/bin/bash
# round 3, call H: conv_wino2d_kernel after the load-pipeline fixes: conv_bench + step-level A/B (wino2d 1 vs 0, alternating)
O=gpurun_out/r03h
mkdir -p $O
timeout 600 tools/bin/conv_bench 5 -1 "wino2d q8 8x64 f68,wino2d q8 8x32 f68,wino2d q16 4x64 f68,wino2d q16 4x32 f68,wino2d q8 8x32 f1028,wino43 q8 8x32x64 t2x1 f32772,wino43 q8 nh1 8x32x64 t1x1 f32772,wino43 q16 nh1 4x64x64 t1x1 f32772,wino43 q8 8x32x32 t1x1 f32772" > $O/conv_bench_w2d.log 2>&1
grep -E "^==|TF/s" $O/conv_bench_w2d.log
for L in 1 0 1 0; do
  timeout 600 python bench.py --no-cpu-baseline --no-split --wino2d $L --steps 20 --profile-out $O/per_op_w2d$L.json > $O/bench_w2d$L.json 2> $O/bench.err
  echo "wino2d $L: $(python -c "import json;d=json.load(open('$O/bench_w2d$L.json'));print(d['ms_per_step'], d['value'], d['roofline']['class_ms_per_step'], d['roofline']['frac'], d['roofline']['direct_equivalent']['ratio_to_peak'])")"
done
