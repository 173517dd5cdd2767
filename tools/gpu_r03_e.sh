#!/bin/bash
# round 3, call E: nested Winograd F(4,3)x x F(2,3)y prototype (conv_wino2d_kernel) vs the 1-D F(4,3) tiles, conv_bench only
O=gpurun_out/r03e
mkdir -p $O
timeout 600 tools/bin/conv_bench 5 -1 "wino 4x64x64 w4x2,wino2d,wino43 q8 8x32x64 t2x1 f32772,wino43 q8 nh1 8x32x64 t1x1 f32772,wino43 q16 nh1 4x64x64 t1x1 f32772,wino43 q8 8x32x32 t1x1 f32772,wino43 q16 4x64x32 t1x1 f65540" > $O/conv_bench_w2d.log 2>&1
echo "conv_bench rc=$?"; cat $O/conv_bench_w2d.log
