#!/bin/bash
# One measurement round on the GPU box (any round: the tag names it); everything lands in gpurun_out/$TAG_* (copy what matters into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh r05_z [tests]'
TAG=${1:-r05_z}; WITH_TESTS=${2:-}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.smoke()' > $O/${TAG}_smoke.log 2>&1; tail -3 $O/${TAG}_smoke.log
if [ -n "$WITH_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=8 > $O/${TAG}_pytest_gpu.log 2>&1
  tail -12 $O/${TAG}_pytest_gpu.log
fi
# bench lines: headline, config[3]'s share (4 lanes x 4) + lock step, config[2] (DDIM-50, batch 16), larger per-GPU shares, DSG+ dims
timeout 400 python bench.py > $O/${TAG}_bench.log 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B --clips-per-gpu 16 --steps 2 --warmup 1 > $O/${TAG}_bench_16clips_l4_b4.log 2>&1
$B --clips-per-gpu 16 --lanes 1 --steps 2 --warmup 1 > $O/${TAG}_bench_16clips_lockstep.log 2>&1
$B --clips-per-gpu 16 --lanes 1 --sampler ddim50 --steps 5 --warmup 1 > $O/${TAG}_bench_ddim50_b16_lockstep.log 2>&1
$B --clips-per-gpu 32 --steps 1 --warmup 1 > $O/${TAG}_bench_32clips_l4_b8.log 2>&1
$B --clips-per-gpu 48 --steps 1 --warmup 1 > $O/${TAG}_bench_48clips_l4_b12.log 2>&1
$B --clips-per-gpu 64 --steps 1 --warmup 1 > $O/${TAG}_bench_64clips_l4_b16.log 2>&1
$B --clips-per-gpu 128 --steps 1 --warmup 1 > $O/${TAG}_bench_128clips_l4_b32.log 2>&1
$B --clips-per-gpu 192 --steps 1 --warmup 1 > $O/${TAG}_bench_192clips.log 2>&1
$B --clips-per-gpu 256 --steps 1 --warmup 1 > $O/${TAG}_bench_256clips.log 2>&1
$B --clips-per-gpu 64 --lanes 1 --steps 1 --warmup 1 > $O/${TAG}_bench_64clips_lockstep.log 2>&1
$B --config beat --steps 1 > $O/${TAG}_bench_beat.log 2>&1
$B --config twh --steps 1 > $O/${TAG}_bench_twh.log 2>&1
$B --config beat --precision bf16w2 --steps 1 --warmup 0 > $O/${TAG}_bench_beat_bf16w2.log 2>&1
$B --config beat --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${TAG}_bench_beat_16clips_l4_b4.log 2>&1
$B --config twh --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${TAG}_bench_twh_16clips_l4_b4.log 2>&1
for c in beat twh; do      # DSG+ with clips in flight (round 6: ROWS at latent_dim 384 / 512)
  $B --config $c --clips-per-gpu 32 --lanes 4 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_32clips_l4_b8.log 2>&1
  $B --config $c --clips-per-gpu 64 --lanes 4 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_64clips_l4_b16.log 2>&1
  python tools/aql_timeline.py --config $c --batch 16 --kset rows --steps 300 --first 100 --n 16 --out $O/${TAG}_aql_step_timeline_${c}_b16_rows.json > /dev/null 2>&1
done
$B --sub-records off --precision bf16w2 --steps 2 --warmup 1 > $O/${TAG}_bench_bf16w2.log 2>&1
$B --sub-records off --precision bf16w2 --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${TAG}_bench_bf16w2_16clips_lockstep.log 2>&1
$B --sub-records off --precision bf16w2 --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${TAG}_bench_bf16w2_16clips_l4_b4.log 2>&1
$B --sub-records off --precision fp32 --steps 1 --warmup 1 > $O/${TAG}_bench_fp32.log 2>&1
$B --precision fp32 --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${TAG}_bench_fp32_16clips_l4_b4.log 2>&1
timeout 600 python tools/e2e.py --reps 3 > $O/${TAG}_e2e_wav_to_bvh.log 2>&1
python tools/aql_timeline.py --config beat --steps 300 --first 100 --n 16 --out $O/${TAG}_aql_step_timeline_beat.json > /dev/null 2>&1
python tools/aql_timeline.py --config twh --steps 300 --first 100 --n 16 --out $O/${TAG}_aql_step_timeline_twh.json > /dev/null 2>&1
# kernel sets side by side (one process, same inputs)
timeout 900 python tools/sweep.py --steps 150 --reps 3 --spec latency:1x1,tile:1x1,latency:4x1,latency:1x2,tile:1x2,tile:4x2,tile:1x4,block:1x4,block:1x8,rows:1x8,rows:4x4,block:4x4,rows:1x16,block:1x16,tile:1x16,rows:4x8,stream:4x8,rows:4x12,stream:4x12,rows:1x24,stream:1x24,rows:1x46,stream:1x48,stream:4x16,stream:1x32,rows:1x32,stream:4x32,stream:1x64,stream:4x64 2>&1 | grep -v amdgpu.ids > $O/${TAG}_sweep_kernel_sets.log
# the timed path's own timeline (in-kernel stamps; command-processor timestamps)
python tools/aql_timeline.py --out $O/${TAG}_aql_step_timeline.json > $O/${TAG}_aql_step_timeline.log 2>&1
python tools/aql_timeline.py --batch 16 --n 16 --out $O/${TAG}_aql_step_timeline_b16.json > /dev/null 2>&1
python tools/aql_timeline.py --batch 16 --precision bf16w2 --n 16 --out $O/${TAG}_aql_step_timeline_b16_bf16w2.json > /dev/null 2>&1
python tools/aql_timeline.py --lib product --out $O/${TAG}_aql_step_timeline_cp_timestamps.json > /dev/null 2>&1
python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/${TAG}_aql_step_timeline_b64_stream.json > /dev/null 2>&1
# PMC traffic of the batched sets the sub-records of bench.py cite (round 6)
bash tools/measure_traffic.sh $TAG 16 rows > $O/${TAG}_traffic_b16.txt 2>&1
bash tools/measure_traffic.sh $TAG 4 rows > $O/${TAG}_traffic_b4.txt 2>&1
bash tools/measure_traffic.sh $TAG 64 stream 20 > $O/${TAG}_traffic_b64.txt 2>&1
# rocprofv3 (HIP-launch path): kernel stats at batch 1 / 16 / 64, PMC traffic (separate passes), MFMA / SQ counters
bash tools/prof.sh ${TAG}_b1 latency:1x1:hip 100 > $O/${TAG}_prof_b1.txt 2>&1
bash tools/prof.sh ${TAG}_b16 rows:1x16:hip 50 > $O/${TAG}_prof_b16.txt 2>&1
bash tools/prof.sh ${TAG}_b64_stream stream:1x64:hip 30 > $O/${TAG}_prof_b64_stream.txt 2>&1
bash tools/prof.sh ${TAG}_beat_b16_rows rows:1x16:hip 30 --config beat > $O/${TAG}_prof_beat_b16_rows.txt 2>&1
bash tools/prof.sh ${TAG}_twh_b16_rows rows:1x16:hip 30 --config twh > $O/${TAG}_prof_twh_b16_rows.txt 2>&1
rm -rf $O/pmc_f_$TAG $O/pmc_w_$TAG
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f_$TAG $O/pmc_w_$TAG 100 > $O/${TAG}_traffic_zeggs_b1_bf16.json 2>$O/${TAG}_traffic.err
bash tools/pmc_mfma.sh > $O/${TAG}_pmc_mfma_b1_b16.log 2>&1
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq_b1_b16.log 2>&1
find $O -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -delete 2>/dev/null
for f in $O/${TAG}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j.get("kernel_set"), j["roofline"]["bound"], j["roofline"]["frac"],
                                 "config3:", (j.get("config3") or {}).get("value"))
PY
done
