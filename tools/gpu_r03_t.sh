#!/bin/bash
# round 3, call T: conv_wino2d_kernel block order: plain (f409664 / f278592), XCD (+4), groups of 256 / 128 / 512 patches (+8 / +16 / +32)
O=gpurun_out/r03t
mkdir -p $O
for s in 0 7 11 6 5 8 3; do timeout 300 tools/bin/conv_bench 5 $s "f409664,f409668,f409672,f409680,f409696,f278592,f278596,f278600,f278608,f278624" ; done > $O/conv_bench_w2d_grp.log 2>&1
echo rc=$?; cat $O/conv_bench_w2d_grp.log
