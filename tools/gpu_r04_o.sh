#!/bin/bash
# round 4, call O: packed fp32 math (v_pk_fma_f32 / v_pk_add_f32) for the fragment-side transforms
O=gpurun_out/r04o
mkdir -p $O
timeout 600 tools/bin/w2d_bench 5 -1 "r3  8x64_RM,r3  8x32_R,w43 q8 8x64 n1,w2d 64,w2d 32" > $O/w2d_bench.log 2>&1; echo "rc=$?"
grep -v "abl-\|plain\|time" $O/w2d_bench.log | cut -c1-175
