# which kernel set per batch size (ZEGGS, bf16, DDPM step, default AQL submission)
for B in 1 2 3 4 6; do
  for lat in on off; do
    echo -n "B=$B latency=$lat: "; timeout 120 python tools/step_timing.py --batch $B --steps 100 --reps 3 --latency $lat 2>&1 | tail -1 | sed 's/.*rep2: //' | cut -d' ' -f1-2
  done
done
