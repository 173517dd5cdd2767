#!/bin/bash
# round 3, call T: raw LDS staging (flags 147524): timing ablations
O=gpurun_out/r03t
mkdir -p $O
for s in 0 11; do timeout 300 tools/bin/conv_bench 5 $s "f147524,f147780,f148036,f148548,f149572,f180292,f213060" ; done > $O/conv_bench_w2d_raw_abl.log 2>&1
echo rc=$?; cat $O/conv_bench_w2d_raw_abl.log
