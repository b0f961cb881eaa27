#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_w
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "stream or STREAM or ffn or rows_do_not" > $O/${T}_pytest_stream.log 2>&1; tail -3 $O/${T}_pytest_stream.log
B="timeout 300 python bench.py --no-cpu-baseline"
for n in 48 64 128 192 256; do $B --clips-per-gpu $n --steps 1 --warmup 1 > $O/${T}_bench_${n}clips.log 2>&1; done
for f in $O/${T}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["config"]["lanes"], "x", j["config"]["batch_per_lane"], j.get("kernel_set"), j["roofline"]["frac"])
PY
done
python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/${T}_aql_step_timeline_b64_stream.json 2>&1 | grep -E "^ *[0-9]+ " | sed -n 1,6p
timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x20,stream:1x20,block:1x24,stream:1x24,block:4x8,stream:4x8,block:4x12,stream:4x12,block:4x10,stream:4x10 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_thresholds.log
