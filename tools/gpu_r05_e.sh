#!/bin/bash
# round 5, call E: the hipGraphLaunch crash of the full-suite order - variants as concurrent processes, staggered (control with the life-cycle
# log, stream levelling around the instantiate, single-lane graphs, engine streams never destroyed)
R=$PWD; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R; rm -f $O/*
run() { name=$1; shift; ( env "$@" FILM_DBG_LOG=$O/dbg_$name.txt SEGV_BT_OUT=$O/segv_$name.txt LD_PRELOAD=$R/tools/bin/segv_bt.so timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -p no:faulthandler -p no:cacheprovider --basetemp=/tmp/pt_$name > $O/log_$name.txt 2>&1; echo "$name rc=$?" >> $O/summary.txt ) & }
run ctl A=1; sleep 25
run lvl FILM_DBG_LEVEL_STREAMS=16; sleep 25
run lan0 FILM_DBG_LANES0=1; sleep 25
run keep FILM_DBG_KEEP_STREAMS=1
wait
cat $O/summary.txt
for f in $O/log_*.txt; do echo "== $f"; grep -v "^\[W\|amdgpu.ids" $f | tail -2 | cut -c1-200; done
head -3 $O/segv_*.txt 2>/dev/null | cut -c1-160
for f in $O/dbg_*.txt; do echo "== $f"; wc -l $f; tail -4 $f; done
