#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "first_layer or tiny or published_64 or config2_256 or batch_and_rect" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-split 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
