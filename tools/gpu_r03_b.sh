#!/bin/bash
# round 3, call B: conv_bench on the F(4,3) variants (Q8 tiles, persistent launches)
O=gpurun_out/r03b
mkdir -p $O
timeout 900 tools/bin/conv_bench 5 -1 wino43 > $O/conv_bench_w43.log 2>&1
echo "conv_bench rc=$?"; grep -c TF $O/conv_bench_w43.log
