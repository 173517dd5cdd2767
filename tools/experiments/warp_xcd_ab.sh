#!/bin/bash
# A/B of the warp kernel's workgroup -> XCD mapping (FILM_WARP_XCD=0: launch order, 1: eight contiguous runs)
mkdir -p gpurun_out
for v in 0 1 0 1; do
  FILM_WARP_XCD=$v timeout 600 python bench.py --no-cpu-baseline --no-split 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('xcd_runs=$v', 'ms_per_step', d['ms_per_step'], 'warp', json.dumps(d['roofline_warp']), 'kernel_ms', json.dumps(d['kernel_ms_per_step']))
" | tee -a gpurun_out/warp_xcd_ab.log
done
