#!/bin/bash
mkdir -p gpurun_out/r04y; cd /root/repo; O=gpurun_out/r04y
for pl in 1 0 1 0; do
timeout 600 python bench.py --no-cpu-baseline --no-split --opt planar=$pl --profile-out $O/prof_$pl.json > $O/bench_$pl.json 2> $O/bench_$pl.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04y/bench_$pl.json').read().strip().splitlines()[-1]); print('planar $pl', d['ms_per_step'], d['kernel_ms_per_step'])
a=json.load(open('gpurun_out/r04y/prof_$pl.json'))
print('   ', ' '.join(f"{o['tag'].split(':')[0]}:{o['ms']:.3f}" for o in a['ops'] if o['tag'].endswith('_1') and o['tag'].startswith('fusion')))
PY
done
