#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "256 or published_64 or batch_equals or autotuned or graph_replay" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1; done
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 --profile-out gpurun_out/prof256b.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/prof256b.json'))
for o in a['ops']:
    if o['tag'].startswith(('flow_l3:predict_flow/flow_predictor_shared/conv_0','fusion_l3:fusion/convs_3_1')): print(o['tag'], round(o['ms']*1000,1), 'us tile', o['tile'])
PY
