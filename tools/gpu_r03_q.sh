#!/bin/bash
O=gpurun_out/r03q
mkdir -p $O
for T in 0 4 0 4; do
  FILM_TUNE_MS=$T timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 > $O/bench_t$T.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_t$T.json'));print('tune_ms $T:', d['ms_per_step'], d['value'], d['roofline']['class_ms_per_step'])"
done
for L in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-split --steps 20 --lanes $L > $O/bench_l$L.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench_l$L.json'));print('lanes $L:', d['ms_per_step'], d['value'])"; done
