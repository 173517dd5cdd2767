#!/bin/bash
# round 5, call A: the new parity tests + the self-checking bench line on the default library
R=$PWD; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -k "1080p or graph_replay or device_path or photos or vimeo or second_weight" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/tests.log | grep -i "passed\|failed\|error\|1080p\|HIP runtime\|done:" | tail -20
timeout 900 python bench.py --profile-out $O/per_op_profile.json > $O/bench_1gpu.json 2> $O/bench_1gpu.err
echo "bench rc=$?"; python - <<'P'
import json,sys
try:
    r=json.load(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r05a/bench_1gpu.json'))
    print({k:r[k] for k in ('value','ms_per_step','parity','timed_output_bit_identical_to_first_call','build')})
    print(r['roofline']['frac'], r['roofline']['traffic'], r['roofline']['traffic_note'])
    print(r['roofline']['dominant_kernel'])
    print(r.get('host_buffers')); print(r.get('cpu_baseline',{}).get('value'))
except Exception as e: print('no line', e)
P
tail -5 $O/bench_1gpu.err
