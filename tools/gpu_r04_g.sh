#!/bin/bash
# round 4, call G: conv_wino2df_kernel with the new epilogue (swapped MFMA operands, one exchange round per x position, dwordx4 stores)
O=gpurun_out/r04g
mkdir -p $O
timeout 900 tools/bin/w2d_bench 5 > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-" $O/w2d_bench.log | cut -c1-190
