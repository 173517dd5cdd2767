#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for wl in 256 256 vimeo_b8; do timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 30 2>/dev/null | tail -1 | cut -c1-140; done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-split 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['roofline_warp']['frac'], d['roofline_warp']['launches_per_step'])"; done
