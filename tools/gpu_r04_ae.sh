#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04ae
timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 --profile-out gpurun_out/r04ae/prof256.json 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
for px in 1024 4096 8192; do echo "w2d_min_px $px"; timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 --opt w2d_min_px=$px 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
for ln in 0 1; do echo "lanes $ln"; timeout 300 python bench.py --no-cpu-baseline --no-split --workload 256 --steps 50 --lanes $ln 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
