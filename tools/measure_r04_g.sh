#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r04_g}
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1; tail -6 $O/${T}_pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --config beat --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${T}_bench_beat_16clips_l4_b4.log 2>&1
$B --config twh --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${T}_bench_twh_16clips_l4_b4.log 2>&1
$B --config beat --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${T}_bench_beat_16clips_lockstep.log 2>&1
$B --config beat --clips-per-gpu 4 --steps 1 --warmup 1 > $O/${T}_bench_beat_4clips_l4_b1.log 2>&1
$B --config beat --steps 1 --warmup 1 > $O/${T}_bench_beat.log 2>&1
$B --precision fp32 --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${T}_bench_fp32_16clips_l4_b4.log 2>&1
$B --precision fp32 --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${T}_bench_fp32_16clips_lockstep.log 2>&1
$B --precision fp32 --steps 1 --warmup 1 > $O/${T}_bench_fp32.log 2>&1
for f in $O/${T}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j.get("kernel_set"), j["roofline"]["bound"], j["roofline"]["frac"])
PY
done
