#!/bin/bash
TAG=${1:-r02_r}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -5 $O/${TAG}_pytest_gpu.log
timeout 300 python tools/pin_check.py --lanes 1,4 --windows 2 --base-env "DSG_PIN=0,DSG_UC=0" --env "DSG_PIN=0,DSG_UC=1" > $O/${TAG}_uc_check.log 2>&1; grep -v "^$" $O/${TAG}_uc_check.log | tail -10
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"])
PY
}
run c1 "" X=1
run c1_uc0 "" DSG_UC=0
run c16_l4 "--clips-per-gpu 16" X=1
run c16_l4_uc0 "--clips-per-gpu 16" DSG_UC=0
run c64 "--clips-per-gpu 64" X=1
run c64_uc0 "--clips-per-gpu 64" DSG_UC=0
run ddim_b16 "--clips-per-gpu 16 --lanes 1 --sampler ddim50 --steps 5" X=1
run ddim_b16_uc0 "--clips-per-gpu 16 --lanes 1 --sampler ddim50 --steps 5" DSG_UC=0
