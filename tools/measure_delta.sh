#!/bin/bash
# A delta pass after a change that touches one model family only (round 6: the DSG+ kernels after the full pass r06_z): smoke, pytest -m gpu, the default
# bench.py line, the DSG+ bench lines and rocprofv3 kernel stats of the DSG+ ROWS step.   gpurun --timeout 1800 -- 'bash tools/measure_delta.sh r06_zz'
TAG=${1:-r06_zz}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.smoke()' > $O/${TAG}_smoke.log 2>&1; tail -3 $O/${TAG}_smoke.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
timeout 500 python bench.py > $O/${TAG}_bench.log 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
for c in beat twh; do
  $B --config $c --steps 1 > $O/${TAG}_bench_${c}.log 2>&1
  $B --config $c --clips-per-gpu 16 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_16clips_l4_b4.log 2>&1
  $B --config $c --clips-per-gpu 16 --lanes 1 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_16clips_lockstep.log 2>&1
  $B --config $c --clips-per-gpu 32 --lanes 4 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_32clips_l4_b8.log 2>&1
  $B --config $c --clips-per-gpu 64 --lanes 4 --steps 1 --warmup 1 > $O/${TAG}_bench_${c}_64clips_l4_b16.log 2>&1
  bash tools/prof.sh ${TAG}_${c}_b16_rows rows:1x16:hip 30 --config $c > $O/${TAG}_prof_${c}_b16_rows.txt 2>&1
done
find $O -name "*_kernel_trace.csv" -delete 2>/dev/null
for f in $O/${TAG}_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j.get("kernel_set"), j["roofline"]["bound"], j["roofline"]["frac"],
                                 "config3:", (j.get("config3") or {}).get("value"), "beat_64clips:", ((j.get("config4") or {}).get("beat_64clips") or {}).get("value"))
PY
done
