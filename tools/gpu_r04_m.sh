#!/bin/bash
# round 4, call M: level-size threshold of the nested kernel (w2d_min_px) on the small-frame / batched configs
O=gpurun_out/r04m
mkdir -p $O
for wl in vimeo_b8 256 photos 1080p_2x2; do
  for px in 8192 4096 2048 1024; do
    timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps 20 --opt w2d_min_px=$px > $O/bench_${wl}_$px.json 2>> $O/bench.err
    python -c "import json;d=json.load(open('$O/bench_${wl}_$px.json'));print('$wl', $px, d['ms_per_step'], d['kernel_ms_per_step']['conv_mfma'])"
  done
done
