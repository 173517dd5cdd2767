#!/bin/bash
# round 4, call H: the new conv_wino2d_kernel in the engine: nested-kernel GPU tests, then the 1080p bench
O=gpurun_out/r04h
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py -m gpu -x -q -k "nested or default_plan" > $O/tests_nested.log 2>&1; echo "nested tests rc=$?"; tail -3 $O/tests_nested.log
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_warp']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
