#!/bin/bash
# round 5, call H: what the driver runs at round end, on the final tree - pytest -x -q -m gpu, smoke, bench (the PMC json of the final sources is in profiles/: traffic replayed)
R=$PWD; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
timeout 400 python bench.py > $O/r05_bench_1gpu_driver_like.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; r=json.load(open('$O/r05_bench_1gpu_driver_like.json')); print(r['ms_per_step'], r['value'], r['parity']['max_abs_vs_ref_graph_golden'], r['roofline']['frac'], r['roofline']['traffic'], r['roofline_warp']['traffic'], r['build'])"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke:"
timeout 500 python -m pytest tests/ -x -q -m gpu > $O/driver_like_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/driver_like_gpu_tests.log
