#!/bin/bash
# round 4, call B: conv_wino2df_kernel with three stages + DMA requests spread over the MFMA gaps
O=gpurun_out/r04b
mkdir -p $O
timeout 900 tools/bin/w2d_bench 5 > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-" $O/w2d_bench.log | grep "==\|w2d 8x64_RM \|w2d 8x32_R\|w2d 8x32_M \|w2f\|mism"
