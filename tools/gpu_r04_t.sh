#!/bin/bash
# graph-replay race: does it reproduce without the relay edges, are the captured edges complete, and which runtime knob hides it
mkdir -p gpurun_out; cd /root/repo
export RACE_SHAPES=2
run() { echo "=== $*"; env "$@" timeout 300 python tools/graph_race_check.py 3 2>&1 | grep -v "^$" | tail -${TAILN:-14}; }
TAILN=40 run FILM_GRAPH_DEBUG=3
run FILM_GRAPH_DEBUG=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run FILM_GRAPH_DEBUG=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run FILM_GRAPH_DEBUG=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run FILM_GRAPH_DEBUG=1 AMD_SERIALIZE_KERNEL=3
run FILM_GRAPH_DEBUG=1 HIP_LAUNCH_BLOCKING=1
run FILM_GRAPH_DEBUG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run FILM_GRAPH_DEBUG=0
echo "=== f1 pipeline after the decode-ahead edit"
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "uint8 or cli" 2>&1 | tail -3
timeout 600 python bench.py --workload cli_1080p_T3 2>&1 | tail -1
