#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04ac
timeout 600 tools/bin/w2d_bench 3 -1 "w2d 64,w2d 32" 2>&1 | grep -v "^$" | awk '/^==/ {print} /w2d 64   |w2d 32   |w2d 64 xf|w2d 32 xf|w2d 64 time|w2d 32 time|\[time\]|mismatch/ {print}' | cut -c1-260 > gpurun_out/r04ac/w2d_bench.log
tail -3 gpurun_out/r04ac/w2d_bench.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r3.py -q -m gpu -x -k "published_64 or config2_256 or autotuned or every_level or tile or graph_replay or fused_rgb or nested" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-split 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['dominant_kernel']['frac'], [c['frac'] for c in d['roofline']['dominant_kernel']['by_input_channels']])"; done
for wl in 256 vimeo_b8; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 2>/dev/null | tail -1 | cut -c1-140; done
