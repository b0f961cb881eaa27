for round in 1 2 3; do
echo -n "eager: "; timeout 60 python tools/step_timing.py --latency on --spg 0 --reps 3 2>&1 | tail -1 | sed 's/.*rep2: //'
for g in 10 50 250; do echo -n "graph spg=$g capture=0: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 60 python tools/step_timing.py --latency on --spg $g --reps 3 2>&1 | tail -1 | sed 's/.*rep2: //'; done
done
