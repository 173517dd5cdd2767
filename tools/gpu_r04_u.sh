#!/bin/bash
# graph-replay race: (1) the round-2 tree that first needed the relay edges, with and without them, on today's box;
# (2) the current tree without relay edges across the whole regression test + the stand-alone checker
mkdir -p gpurun_out; cd /root/repo
O=tools/bin/old_r2
echo "=== round-2 tree, relay edges OFF"; FILM_NO_RELAY=1 timeout 300 python $O/tools/dbg_graph_race.py 3 2>&1 | tail -13
echo "=== round-2 tree, relay edges ON";  timeout 300 python $O/tools/dbg_graph_race.py 3 2>&1 | tail -13
echo "=== current tree, relay OFF: stand-alone, fuse 3 and 31"
FILM_GRAPH_DEBUG=1 timeout 300 python tools/graph_race_check.py 3 2>&1 | tail -25
FILM_GRAPH_DEBUG=1 timeout 300 python tools/graph_race_check.py 31 2>&1 | tail -25
echo "=== current tree, relay OFF: graph tests"
FILM_GRAPH_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "graph or published_64 or lanes" 2>&1 | tail -5
