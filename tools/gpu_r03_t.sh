#!/bin/bash
# round 3, call T: conv_wino43_kernel PF2 tiles, K loop as unconditional chunk pairs (f32772) vs the odd chunk under a condition (f49156)
O=gpurun_out/r03t
mkdir -p $O
for s in 2 1 12 9 10 13 3; do timeout 300 tools/bin/conv_bench 5 $s "f32772,f49156" ; done > $O/conv_bench_w43_loop.log 2>&1
echo rc=$?; cat $O/conv_bench_w43_loop.log
