#!/bin/bash
# A/B of the STREAM pose head on one box: DSG_WS_OUT_ONE=0|1 (persistent row-block groups / one workgroup per row block, three per CU).
#   gpurun --timeout 600 -- 'bash tools/ab_pose_head.sh r05_x'
TAG=${1:-r05_x}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="timeout 300 python bench.py --no-cpu-baseline --sub-records off"
for E in 0 1 0 1; do
  export DSG_WS_OUT_ONE=$E
  python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/${TAG}_timeline_b64_stream_one$E.json > /dev/null 2>&1
  python - $O/${TAG}_timeline_b64_stream_one$E.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["us_per_step_untraced_runs"], {k: (v["launches_per_step"], round(v["busy_us"], 2)) for k, v in d["by_kernel"].items() if k.startswith("k_ws<")})
PY
done
for E in 0 1; do
  export DSG_WS_OUT_ONE=$E
  $B --clips-per-gpu 64 --lanes 1 --steps 1 --warmup 1 > $O/${TAG}_bench_64clips_lockstep_one$E.log 2>&1
  $B --clips-per-gpu 256 --steps 1 --warmup 1 > $O/${TAG}_bench_256clips_one$E.log 2>&1
done
unset DSG_WS_OUT_ONE
for f in $O/${TAG}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j.get("kernel_set"))
PY
done
timeout 300 python -m pytest tests/test_gpu_round5.py -m gpu -q -p no:cacheprovider -k "one_block" 2>&1 | tail -4
