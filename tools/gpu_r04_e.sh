#!/bin/bash
# round 4, call E: two channel tiles per wave (64 channels, 192 accumulators, one wave per SIMD)
O=gpurun_out/r04e
mkdir -p $O
timeout 900 tools/bin/w2d_bench 5 -1 "w2d 8x64_RM,w2f 64 ns3,w2f 32 ns3,n2" > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep "==\|w2d\|ns3  \|ns3 yf\|n2\|mism" $O/w2d_bench.log | cut -c1-118
