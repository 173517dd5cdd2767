#!/bin/bash
# round 5, call D: native backtrace of the segfault in test_4k_frame_crosses_the_4gib_buffer_rule (full-suite order)
R=$PWD; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
SEGV_BT_OUT=$O/segv.txt LD_PRELOAD=$R/tools/bin/segv_bt.so timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:faulthandler > $O/bt.log 2>&1
echo "rc=$?"
cat $O/segv.txt | cut -c1-200
