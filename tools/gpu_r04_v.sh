#!/bin/bash
# graph race: the stand-alone reduction (chain / fan, 0..8 waiters, relay) + the engine with every cross-lane wait emitted once
mkdir -p gpurun_out; cd /root/repo
timeout 200 tools/bin/graph_single_parent_race
echo "=== DEBUG_HIP_FORCE_GRAPH_QUEUES=8"; DEBUG_HIP_FORCE_GRAPH_QUEUES=8 timeout 200 tools/bin/graph_single_parent_race | grep "relay 0" | grep "spin  200"
echo "=== engine"
timeout 300 python tools/graph_race_check.py 3 2>&1 | awk '{print $1, $2, $3, $4, $5, $NF}' | sort | uniq -c | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "graph or published_64 or lanes" 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1
