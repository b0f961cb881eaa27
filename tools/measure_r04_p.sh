#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_p
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider -k "block or BLOCK or sets or arrangement or rows_do_not" > $O/${T}_pytest_block.log 2>&1; tail -3 $O/${T}_pytest_block.log
B="timeout 300 python bench.py --no-cpu-baseline"
for v in 0 1 2 3; do
  DSG_FFN_SPLIT=$v $B --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${T}_bench_16_lockstep_split$v.log 2>&1
  DSG_FFN_SPLIT=$v $B --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${T}_bench_16_lanes_split$v.log 2>&1
  DSG_FFN_SPLIT=$v DSG_KSET=3 $B --clips-per-gpu 32 --lanes 1 --steps 1 --warmup 1 > $O/${T}_bench_32_lockstep_split$v.log 2>&1
done
for f in $O/${T}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["config"]["lanes"], "x", j["config"]["batch_per_lane"], j.get("kernel_set"), j["roofline"]["frac"])
PY
done
for v in 1 3; do DSG_FFN_SPLIT=$v python tools/aql_timeline.py --batch 16 --kset block --steps 120 --first 40 --n 16 --out $O/${T}_timeline_b16_split$v.json 2>&1 | grep -E "^ *[0-9]+ " | sed -n 3,12p; done
