#!/bin/bash
# last evidence refresh of round 4: full GPU suite, smoke, the 1080p and 256x256 bench lines of the final tree
R=$PWD; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -s > $O/r04_gpu_tests_full.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $O/r04_gpu_tests_full.log | tail -2
timeout 300 python __graft_entry__.py smoke > $O/r04_smoke.log 2>&1; echo "smoke rc=$?"; grep -i "smoke:" $O/r04_smoke.log | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 30 > $O/r04_bench_256.json 2>> $O/bench.err; cut -c1-170 $O/r04_bench_256.json
timeout 900 python bench.py --profile-out $O/r04_per_op_profile.json > $O/r04_bench_1gpu.json 2>> $O/bench.err; cut -c1-200 $O/r04_bench_1gpu.json
