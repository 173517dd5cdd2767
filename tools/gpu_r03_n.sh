#!/bin/bash
# the library now really contains the fixed conv_wino2d_kernel: parity subset + step-level A/B
O=gpurun_out/r03n
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_r3.py tests/test_gpu_configs.py -m gpu -x -q -s -k "nested or default_plan_uses or tile_960x576 or 1080p_2x2 or depth6" > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_tests.log | tail -4 | cut -c1-300
for L in 1 0 1 0; do
  timeout 600 python bench.py --no-cpu-baseline --no-split --wino2d $L --steps 20 --profile-out $O/per_op_w2d$L.json > $O/bench_w2d$L.json 2> $O/bench.err
  echo "wino2d $L: $(python -c "import json;d=json.load(open('$O/bench_w2d$L.json'));print(d['ms_per_step'], d['value'], d['roofline']['class_ms_per_step'], d['roofline']['frac'], d['roofline']['direct_equivalent']['ratio_to_peak'])")"
done
python tools/prof_compare.py $O/per_op_w2d1.json $O/per_op_w2d0.json 200 | grep -E "tile 8[0-9]{3} |conv total"
