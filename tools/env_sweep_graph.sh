run() { echo -n "$1 spg=$2: "; timeout 60 env $1 python tools/step_timing.py --latency on --spg $2 --reps 3 2>&1 | tail -1 | sed 's/.*rep2: //' | cut -d' ' -f1-2; echo; }
run X=0 0
run X=0 50
for e in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=256 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 HIP_FORCE_DEV_KERNARG=0; do run $e 50; done
