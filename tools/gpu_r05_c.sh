#!/bin/bash
# round 5, call C (re-entry): the full GPU suite of the restored tree with wall-clock per step, 1080p bench + per-op profile, 256 / vimeo lines
R=$PWD; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
t0=$(date +%s); lap() { t1=$(date +%s); echo "[$1: $((t1-t0)) s]"; t0=$t1; }
timeout 1500 python -m pytest tests -m gpu -q -s -x > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/gpu_tests.log | grep -i "passed\|failed\|error" | tail -5; lap tests
timeout 600 python bench.py --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo "bench rc=$?"; lap bench
for wl in 256 vimeo_b8; do timeout 200 python bench.py --no-cpu-baseline --workload $wl --steps 30 --profile-out $O/per_op_profile_$wl.json > $O/bench_$wl.json 2>> $O/bench_1gpu.err; done; lap small
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05c/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r.get('parity'), r['roofline']['frac'], r['roofline'].get('traffic'))
    except Exception as e: print(f, 'no line', e)
P
tail -3 $O/bench_1gpu.err
