#!/bin/bash
mkdir -p gpurun_out/r04x; cd /root/repo; O=gpurun_out/r04x
timeout 200 tools/bin/warp_bench
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "warp_corner or published_64 or config2_256 or graph_replay or aux or stages" 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-split --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04x/bench_1gpu.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline_warp']['class_ms_per_step'], d['roofline_warp']['frac'], d['kernel_ms_per_step'])
a=json.load(open('gpurun_out/r04x/per_op_profile.json'))
for o in a['ops']:
    if o['kind']=='warp': print(f"{o['tag']:36s} {o['ms']:.4f} ms {o['bytes']/1e6:8.1f} MB {o['bytes']/o['ms']/1e9:.2f} TB/s")
PY
