#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "warp_corner or published_64 or config2_256 or graph_replay" 2>&1 | tail -8
timeout 300 python bench.py 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline_warp'], d['kernel_ms_per_step'])"
