#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_y2
timeout 400 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -p no:cacheprovider -k "ffn_64_row or fused_ffn_vs_oracle_and_lanes" > $O/${T}_pytest.log 2>&1; tail -3 $O/${T}_pytest.log
B="timeout 200 python bench.py --no-cpu-baseline"
for n in 128 192 256; do $B --clips-per-gpu $n --steps 1 --warmup 1 > $O/${T}_bench_${n}clips.log 2>&1; done
for f in $O/${T}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["config"]["lanes"], "x", j["config"]["batch_per_lane"], j.get("kernel_set"), j["roofline"]["frac"])
PY
done
