#!/bin/bash
# One full measurement round on the GPU box (round 2); everything lands in gpurun_out/$TAG_* (copy what matters into profiles/).
#   gpurun --timeout 2400 -- 'bash tools/measure_round2.sh r02_z'
TAG=${1:-r02_z}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1
tail -3 $O/${TAG}_pytest_gpu.log
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B --clips-per-gpu 16 --steps 2 --warmup 1 > $O/${TAG}_bench_16clips_l4_b4.log 2>&1
$B --clips-per-gpu 16 --lanes 1 --steps 2 --warmup 1 > $O/${TAG}_bench_16clips_lockstep.log 2>&1
$B --clips-per-gpu 16 --mode streams --steps 1 --warmup 1 > $O/${TAG}_bench_16clips_streams16.log 2>&1
$B --clips-per-gpu 4 --steps 2 --warmup 1 > $O/${TAG}_bench_4clips_l4_b1.log 2>&1
$B --clips-per-gpu 64 --steps 1 --warmup 1 > $O/${TAG}_bench_64clips_l4_b16.log 2>&1
$B --clips-per-gpu 128 --steps 1 --warmup 1 > $O/${TAG}_bench_128clips_l4_b32.log 2>&1
$B --clips-per-gpu 16 --lanes 1 --sampler ddim50 --steps 5 --warmup 1 > $O/${TAG}_bench_ddim50_b16_lockstep.log 2>&1
$B --clips-per-gpu 16 --sampler ddim50 --steps 5 --warmup 1 > $O/${TAG}_bench_ddim50_16clips_l4_b4.log 2>&1
$B --config beat --steps 1 > $O/${TAG}_bench_beat.log 2>&1
$B --config twh --steps 1 > $O/${TAG}_bench_twh.log 2>&1
python tools/step_timing.py --latency on,off --reps 3 > $O/${TAG}_step_timing.log 2>&1
DSG_AQL=0 python tools/step_timing.py --latency on --reps 3 > $O/${TAG}_step_timing_hip_launches.log 2>&1
DSG_LIB=diffusestylegesture_amd/csrc/libdsg_hip_stamps.so python tools/stamps.py > $O/${TAG}_stamps.log 2>&1
rm -rf $O/prof_$TAG $O/pmc_f_$TAG $O/pmc_w_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o z -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess > $O/${TAG}_prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f_$TAG $O/pmc_w_$TAG 100 > $O/${TAG}_traffic_zeggs_b1_bf16.json 2>$O/${TAG}_traffic.err
find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
bash tools/prof_batch.sh 16 $TAG > /dev/null 2>&1
bash tools/pmc_mfma.sh > $O/${TAG}_pmc_mfma_b1_b16.log 2>&1
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq_b1_b16.log 2>&1
# the per-dispatch traces are large; keep only the summaries
find $O -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -delete 2>/dev/null
for f in $O/${TAG}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j["roofline"]["bound"], j["roofline"]["frac"], "post", j.get("postprocess_ms_per_clip"), "e2e", j.get("value_end_to_end"))
PY
done
