#!/bin/bash
# round 4, call Q: engine bench after packed math + threshold 1536 (all BASELINE configs), per-op profile
O=gpurun_out/r04q
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --profile-out $O/per_op_profile.json > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'], d['roofline_warp']['frac'])"
for wl in vimeo_b8 256 photos; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 > $O/bench_$wl.json 2>> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', d['ms_per_step'], d['kernel_ms_per_step']['conv_mfma'])"; done
