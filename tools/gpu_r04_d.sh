#!/bin/bash
# round 4, call D: y-first transform (72 VALU per chunk instead of 120) + the 12-operation x transform
O=gpurun_out/r04d
mkdir -p $O
timeout 900 tools/bin/w2d_bench 5 -1 "w2d 8x64_RM,w2d 8x32_R,w2f 64 ns3,w2f 32 ns3,yf" > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-" $O/w2d_bench.log | cut -c1-118
