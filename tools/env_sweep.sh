# runtime-knob sweep (measured r01: AMD_OPT_FLUSH=0 -> 217 us/step vs 151 default, i.e. system-scope fences cost ~2 us per launch)
# runtime-knob sweep for the batch-1 step (same box, same build): eager and hipGraph replay
# every run is wrapped in `timeout`: makes the host miss the completion signal and hangs forever
run() { echo -n "$1 spg=$2: "; timeout 60 env $1 python tools/step_timing.py --latency on --spg $2 --reps 3 2>&1 | tail -1 | sed 's/.*rep2: //'; }
run X=0 0
for e in AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 AMD_DIRECT_DISPATCH=0 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 GPU_FLUSH_ON_EXECUTION=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 ROC_ACTIVE_WAIT_TIMEOUT=0; do run $e 0; done
run X=0 0
run X=0 50
for e in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=64 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 AMD_OPT_FLUSH=0; do run $e 50; done
