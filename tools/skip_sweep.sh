for m in 0 1 2 4 8 16 32; do echo -n "skip=$m "; DSG_DEBUG_SKIP=$m python tools/step_timing.py --latency on 2>&1 | tail -1; done
