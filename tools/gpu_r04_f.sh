#!/bin/bash
# round 4, call F: where a workgroup's time goes (s_memtime at entry / first MFMA / last MFMA / exit)
O=gpurun_out/r04f
mkdir -p $O
timeout 900 tools/bin/w2d_bench 3 -1 "time" > $O/w2d_time.log 2>&1; echo "rc=$?"
cat $O/w2d_time.log | cut -c1-200
