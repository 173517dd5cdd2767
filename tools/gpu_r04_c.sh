#!/bin/bash
# round 4, call C: where do conv_wino2df_kernel's cycles go - clock, MFMA busy, stall split (PMC), fusion_1_1 and flow_l3_c0
for sh in 0 2; do
  bash tools/gpu_pmc_w2d.sh gpurun_out/r04c/s$sh $sh "w2d 8x64_RM,w2d 8x32_R,w2f 64 ns3,w2f 32 ns3"
done
for sh in 0 2; do echo "== shape $sh"; cat gpurun_out/r04c/s$sh/sq1_table.txt; done
