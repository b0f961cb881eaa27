# A/B several library builds on the same box: tools/ab.sh lib1.so lib2.so ...   (two interleaved rounds each)
for round in 1 2; do
for lib in "$@"; do echo -n "$lib: "; DSG_LIB=$lib python tools/step_timing.py --latency on --reps 4 2>&1 | tail -1 | sed 's/.*rep3: //'; done
done
