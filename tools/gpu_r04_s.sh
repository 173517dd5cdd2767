#!/bin/bash
# round 4, call S: f1 pipeline test + files->files bench + default bench line with host_buffers
O=gpurun_out/r04s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "uint8 or cli" > $O/tests.log 2>&1; echo "tests rc=$?"; grep "passed\|failed\|Error" $O/tests.log | tail -5
timeout 600 python bench.py --workload cli_1080p_T3 --steps 5 --warmup 1 > $O/bench_cli.json 2> $O/bench_cli.err; echo "cli rc=$?"; cat $O/bench_cli.json | cut -c1-1500; tail -3 $O/bench_cli.err
timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_1.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$O/bench_1.json'));print(d['ms_per_step'], d['value'], d.get('host_buffers'), d['roofline']['dominant_kernel'])"
