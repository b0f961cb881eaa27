#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_q
B="timeout 300 python bench.py --no-cpu-baseline"
run() { # name, env...
  n=$1; shift
  env "$@" $B --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${T}_bench_16_lockstep_$n.log 2>&1
  env "$@" $B --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${T}_bench_16_lanes_$n.log 2>&1
  env "$@" $B --clips-per-gpu 32 --steps 1 --warmup 1 > $O/${T}_bench_32_lanes_$n.log 2>&1
}
run s0 DSG_FFN_SPLIT=0
run s2 DSG_FFN_SPLIT=2
run s2rw8 DSG_FFN_SPLIT=2 DSG_FFN_LN_RW=8
run s2rw4 DSG_FFN_SPLIT=2 DSG_FFN_LN_RW=4
run s2last DSG_FFN_SPLIT=2 DSG_FFN_SPLIT_LAST=1
run s0b DSG_FFN_SPLIT=0
run s2b DSG_FFN_SPLIT=2
for f in $O/${T}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["config"]["lanes"], "x", j["config"]["batch_per_lane"], j.get("kernel_set"), j["roofline"]["frac"])
PY
done
DSG_FFN_SPLIT=2 DSG_FFN_SPLIT_LAST=1 python tools/aql_timeline.py --batch 16 --kset block --steps 120 --first 40 --n 16 --out $O/${T}_timeline_b16_s2last.json 2>&1 | grep -E "^ *[0-9]+ " | sed -n 28,36p
