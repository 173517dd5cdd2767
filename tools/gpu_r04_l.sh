#!/bin/bash
# round 4, call L: conv_buf_kernel with swapped MFMA operands + dwordx4 stores; prologue DMA order; sanity + bench
O=gpurun_out/r04l
mkdir -p $O
timeout 300 tools/bin/w2d_bench 3 -1 "r3  8x64_RM,r3  8x32_R,w2d 64,w2d 32" > $O/w2d_bench.log 2>&1; echo "w2d_bench rc=$?"; grep "mismatch" $O/w2d_bench.log
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_warp']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep "passed\|failed" $O/tests.log | tail -3
