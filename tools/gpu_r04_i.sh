#!/bin/bash
# round 4, call I: the nested kernel against the 1-D F(4,3) tiles on the short-K layers the 1-D kernel still runs
O=gpurun_out/r04i
mkdir -p $O
timeout 900 tools/bin/w2d_bench 5 -1 "w43,w2d 64,w2d 32" > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-\|plain\| xf" $O/w2d_bench.log | cut -c1-150
