# A/B two builds (library + code object directories) on the same box:  tools/ab_lib.sh dirA dirB
for round in 1 2 3; do for d in "$@"; do echo -n "$d: "; DSG_LIB=$d/libdsg_hip.so timeout 60 python tools/step_timing.py --latency on --reps 4 2>&1 | tail -1 | sed 's/.*rep3: //'; done; done
