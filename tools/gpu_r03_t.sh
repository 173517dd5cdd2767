#!/bin/bash
# round 3, call T: conv_wino2d_kernel with three activation stages and the barrier behind nu step 3 (MIDBAR = +524288): 786496/786500 register staging, 917568/917572 raw staging
O=gpurun_out/r03t
mkdir -p $O
for s in 4 0 7 11 6 5 8 3; do timeout 300 tools/bin/conv_bench 5 $s "f278592,f278596,f409664,f409668,f786496,f786500,f917568,f917572" ; done > $O/conv_bench_w2d_midbar.log 2>&1
echo rc=$?; cat $O/conv_bench_w2d_midbar.log
