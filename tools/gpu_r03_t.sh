#!/bin/bash
# round 3, call T: weight slabs two chunks ahead (B2 = +262144), with register staging (278596) and raw LDS staging (409668)
O=gpurun_out/r03t
mkdir -p $O
for s in 4 0 7 11 6 5 8; do timeout 300 tools/bin/conv_bench 5 $s "f16452,f147524,f278596,f409668" ; done > $O/conv_bench_w2d_b2.log 2>&1
echo rc=$?; cat $O/conv_bench_w2d_b2.log
