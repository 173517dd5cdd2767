#!/bin/bash
# round 4, call K: fused 1x1 with float4 tile + weights in LDS; full GPU suite; 1080p bench
O=gpurun_out/r04k
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_warp']['frac'])"
python - <<'P'
import json
for o in json.load(open('gpurun_out/r04k/per_op_profile.json'))['ops']:
    if 'output_conv' in o['tag'] or 'convs_0_2' in o['tag']: print(o['tag'], o['ms'])
P
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep "passed\|failed" $O/tests.log | tail -3
