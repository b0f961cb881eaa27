timeout 120 python -m pytest tests -m gpu -x -q -k "aql or batch1" 2>&1 | tail -3
for round in 1 2; do
for v in "DSG_OVERLAP=0" "DSG_OVERLAP=1" "DSG_AQL=0 DSG_OVERLAP=1"; do echo -n "$v: "; timeout 60 env $v python tools/step_timing.py --latency on --reps 4 2>&1 | tail -1 | sed 's/.*rep3: //'; done
done
