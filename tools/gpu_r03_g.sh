#!/bin/bash
# round 3, call G: conv_wino2d_kernel in the engine: parity + bench with / without
O=gpurun_out/r03g
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py -m gpu -x -q -s -k "nested or default_plan_uses" > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_tests.log | tail -12
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench.err
echo "bench: $(python -c "import json;d=json.load(open('$O/bench_1gpu.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['class_ms_per_step'], d['kernel_ms_per_step'])")"; tail -2 $O/bench.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/r03g/per_op_profile.json'))
for o in a['ops']:
    if o['kind']=='conv_mfma' and o['tile'] & 8192: print(f"{o['ms']:.3f} ms tile {o['tile']&15}{'x' if o['tile']&16 else ''} {o['tag']}")
PY
