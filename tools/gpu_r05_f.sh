#!/bin/bash
# round 5, call F: the full GPU suite with the direct two-lane executor as the default, smoke, and the step time of the three
# executors (direct two lanes = default, hipGraph replay, one stream) on the 1080p / 256x256 / vimeo_b8 workloads
R=$PWD; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
t0=$(date +%s); lap() { t1=$(date +%s); echo "[$1: $((t1-t0)) s]"; t0=$t1; }
SEGV_BT_OUT=$O/segv.txt LD_PRELOAD=$R/tools/bin/segv_bt.so timeout 1300 python -m pytest tests -m gpu -q -s -p no:faulthandler > $O/r05_gpu_tests.log 2>&1
echo "tests rc=$?"; grep -v "^\[W\|amdgpu.ids" $O/r05_gpu_tests.log | grep -i "passed\|failed\|error" | tail -8; head -5 $O/segv.txt 2>/dev/null; lap tests
timeout 300 python __graft_entry__.py smoke > $O/r05_smoke.log 2>&1; echo "smoke rc=$?"; grep -i smoke $O/r05_smoke.log | tail -1; lap smoke
for wl in 1080p_2x2 256 vimeo_b8; do
  for mode in "" "--graph" "--no-graph"; do
    st=20; [ $wl = 256 ] && st=100; [ $wl = vimeo_b8 ] && st=40
    timeout 300 python bench.py --no-cpu-baseline --no-split --workload $wl --steps $st $mode > $O/bench_${wl}_exec${mode}.json 2>> $O/bench.err
  done
done; lap benches
python - <<'P' | tee gpurun_out/r05f/r05_bench_exec_modes.log
import json,glob
for f in sorted(glob.glob('gpurun_out/r05f/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['config'].get('exec'), 'ms_per_step', r['ms_per_step'], r['metric'], r['value'], 'parity', (r.get('parity') or {}).get('max_abs_vs_ref_graph_golden'))
    except Exception as e: print(f, 'no line', e)
P
tail -3 $O/bench.err
