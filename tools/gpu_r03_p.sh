#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r3.py -m gpu -x -q -k "tiny_config_stages or published_64 or errors or second_weight or savedmodel or nested_winograd_kernel_on_every_level or rccl" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], [ (k['name'],k['ms'],k['executed_tflops']) for k in d['roofline']['kernels']])"
