# A/B an environment switch on the same box: tools/ab_env.sh VAR val1 val2   (two interleaved rounds)
for round in 1 2; do
for v in "$2" "$3"; do echo -n "$1=$v: "; env $1=$v python tools/step_timing.py --latency on --reps 4 2>&1 | tail -1 | sed 's/.*rep3: //'; done
done
