#!/bin/bash
# round 5, call B: split-K of the nested kernel - per-layer (w2d_bench) and in the forward (bench A/B), parity subset
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
F="=w2d 64,=w2d 32,split"
for sh in 1 2 3 20 21 22; do timeout 120 tools/bin/w2d_bench 5 $sh "$F"; done > $O/w2d_bench_split.log 2>&1
grep -E "^==|w2d|mismatch" $O/w2d_bench_split.log | cut -c1-170
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -k "tile_960 or config2_256 or vimeo or 1080p_2x2_tiled or batch_and_rect" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error" | tail -5
timeout 600 python bench.py --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --opt w2d_splitk=0 --profile-out $O/per_op_profile_nosplit.json > $O/bench_1gpu_nosplit.json 2>> $O/bench_1gpu.err
timeout 300 python bench.py --no-cpu-baseline --profile-out $O/per_op_profile_2.json > $O/bench_1gpu_2.json 2>> $O/bench_1gpu.err
for wl in 256 vimeo_b8; do for o in 1 0; do timeout 200 python bench.py --no-cpu-baseline --workload $wl --steps 30 --opt w2d_splitk=$o > $O/bench_${wl}_split$o.json 2>> $O/bench_1gpu.err; done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r.get('parity'), r.get('timed_output_bit_identical_to_first_call'), r['roofline']['frac'])
    except Exception as e: print(f, 'no line', e)
P
tail -3 $O/bench_1gpu.err
