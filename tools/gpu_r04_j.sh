#!/bin/bash
# round 4, call J: nested kernel on every 3x3 layer of the large levels (fused pool / RGB head in its epilogue): tests + 1080p bench
O=gpurun_out/r04j
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_warp']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_r3.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep "passed\|failed" $O/tests.log | tail -3
