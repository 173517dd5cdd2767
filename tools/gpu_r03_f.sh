#!/bin/bash
O=gpurun_out/r03f
mkdir -p $O
for S in 0 7 11; do timeout 300 tools/bin/conv_bench 5 $S "wino2d,wino43 q8 8x32x64 t2x1 f32772,wino43 q8 8x32x32 t1x1 f32772" ; done > $O/conv_bench_w2d_abl.log 2>&1
cat $O/conv_bench_w2d_abl.log
