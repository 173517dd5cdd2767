#!/bin/bash
# round 5, call I: the crashing sequence (configs + parity in file order, hipGraph executor forced) with 8 hardware queues instead of 4
R=$PWD; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
FILM_TEST_EXECUTOR=1 GPU_MAX_HW_QUEUES=8 SEGV_BT_OUT=$O/segv.txt LD_PRELOAD=$R/tools/bin/segv_bt.so timeout 215 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:faulthandler > $O/log.txt 2>&1
echo "rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/log.txt | tail -3 | cut -c1-200; head -6 $O/segv.txt 2>/dev/null | cut -c1-160
