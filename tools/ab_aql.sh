for round in 1 2; do
for v in "DSG_AQL=0" "DSG_AQL=1" "DSG_AQL=1 DSG_AQL_ACQUIRE=0"; do echo -n "$v: "; env $v python tools/step_timing.py --latency on --reps 4 2>&1 | tail -1 | sed 's/.*rep3: //'; done
done
DSG_AQL_ACQUIRE=0 timeout 200 python -m pytest tests -m gpu -x -q -k "aql or ddpm_chain" 2>&1 | tail -3
