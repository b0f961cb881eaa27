#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_s
for spec in "64 stream b64_stream" "16 block b16_block"; do
  set -- $spec
  rm -rf $O/pmc_f_$T $O/pmc_w_$T
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$T -o z -- python tools/step_timing.py --batch $1 --kset $2 --steps 40 --reps 1 --spg=-1 > $O/${T}_pmc_f_$3.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$T -o z -- python tools/step_timing.py --batch $1 --kset $2 --steps 40 --reps 1 --spg=-1 > $O/${T}_pmc_w_$3.log 2>&1
  python tools/pmc_traffic.py $O/pmc_f_$T $O/pmc_w_$T 40 > $O/${T}_traffic_zeggs_$3_bf16.json 2>$O/${T}_traffic_$3.err
  head -c 400 $O/${T}_traffic_zeggs_$3_bf16.json; echo
done
find $O -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -delete 2>/dev/null
