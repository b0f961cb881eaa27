#!/bin/bash
# HBM-side traffic per denoising step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs; HIP-launch path: the profiler cannot
# see hand-written AQL packets) for one (batch, kernel set[, config]):  bash tools/measure_traffic.sh TAG BATCH KSET [STEPS [CONFIG]]
TAG=$1; B=$2; KSET=$3; STEPS=${4:-40}; CFG=${5:-zeggs}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $O/pmc_f_$TAG $O/pmc_w_$TAG
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$TAG -o z -- python tools/step_timing.py --config $CFG --batch $B --kset $KSET --steps $STEPS --reps 1 --spg=-1 > $O/${TAG}_pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$TAG -o z -- python tools/step_timing.py --config $CFG --batch $B --kset $KSET --steps $STEPS --reps 1 --spg=-1 > $O/${TAG}_pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f_$TAG $O/pmc_w_$TAG $STEPS > $O/${TAG}_traffic_${CFG}_b${B}_${KSET}_bf16.json 2>$O/${TAG}_traffic.err
find $O/pmc_f_$TAG $O/pmc_w_$TAG -name "*.csv" -delete 2>/dev/null
python - $O/${TAG}_traffic_${CFG}_b${B}_${KSET}_bf16.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "fetch raw MB", round(d["fetch_bytes_per_step_raw"] / 1e6, 2), "write MB", round(d["write_bytes_per_step"] / 1e6, 2), "raw", round(d["traffic_bytes_per_step_raw"] / 1e6, 2), "fetch x2", round(d["traffic_bytes_per_step_fetch_x2"] / 1e6, 2))
PY
