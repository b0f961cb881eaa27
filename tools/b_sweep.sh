# which kernel set / workgroup width per batch size (ZEGGS, bf16, DDPM step)
for B in 1 2 3 4 8; do
  for cfgs in "on 1" "off 1" "off 2"; do set -- $cfgs
    echo -n "B=$B latency=$1 TNW=$2: "; DSG_GEMM_TNW=$2 timeout 120 python tools/step_timing.py --batch $B --steps 100 --reps 3 --latency $1 2>&1 | tail -1 | sed 's/.*rep2: //' | cut -d' ' -f1-2
  done
done
