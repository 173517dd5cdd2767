#!/bin/bash
# round 4, call A: conv_wino2df_kernel (fragment-side transforms) against conv_wino2d_kernel's tiles: bit identity + time
O=gpurun_out/r04a
mkdir -p $O
timeout 600 tools/bin/w2d_bench 5 > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-" $O/w2d_bench.log | tail -120
