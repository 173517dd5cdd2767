#!/bin/bash
# 64- vs 32-channel tile of conv_wino2d_kernel: step time and fabric reads with the tile forced to 64 wherever it fits
R=$PWD; O=$R/gpurun_out/r04ab; mkdir -p $O; cd $R
for sh in -1 0 -1 0; do
  timeout 300 python bench.py --no-cpu-baseline --no-split --opt w2d_shape=$sh 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('w2d_shape $sh:', d['ms_per_step'], d['kernel_ms_per_step']['conv_mfma'], d['roofline']['dominant_kernel']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
for sh in -1 0; do
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$sh -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-split --lanes 0 --opt w2d_shape=$sh > $O/fetch_$sh.log 2>&1
  python - <<PY
import csv, glob, collections
f=glob.glob('$O/fetch_$sh/**/*counter_collection.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    k=r['Kernel_Name'].split('(')[0]
    if r['Counter_Name']=='FETCH_SIZE': tot[k]+=float(r['Counter_Value']); n[k]+=1
w=[(k,v) for k,v in tot.items() if 'wino2d' in k]
# two forwards (warmup + step) + autotune launches are in the trace: report the per-launch mean of the LAST 46 launches' share instead
print('w2d_shape $sh: conv_wino2d FETCH_SIZE KB total', sum(v for k,v in w), 'launches', sum(n[k] for k,_ in w))
PY
done
