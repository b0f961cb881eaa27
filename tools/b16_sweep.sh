# batched path: workgroup shape sweep (TM row tiles per workgroup, TNW col tiles per wave) at several batch sizes
for B in 4 8 16; do
for cfgs in "1 1" "4 1" "4 2"; do set -- $cfgs
  echo -n "B=$B TM=$1 TNW=$2: "; DSG_GEMM_TM=$1 DSG_GEMM_TNW=$2 timeout 120 python tools/step_timing.py --batch $B --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //' | cut -d' ' -f1-2
done; done
DSG_GEMM_TM=4 timeout 200 python -m pytest tests -m gpu -q -k "batch16 or throughput or forward_zeggs" 2>&1 | tail -1
