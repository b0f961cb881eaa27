#!/bin/bash
TAG=${1:-r02_t}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1; tail -3 $O/${TAG}_pytest_gpu.log
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"], j.get("fence_free_packets"))
PY
}
run c1 "" X=1
run c1_uc0 "" DSG_UC=0
run c4 "--clips-per-gpu 4" X=1
run c16_l4 "--clips-per-gpu 16" X=1
run c16_l1 "--clips-per-gpu 16 --lanes 1" X=1
run c64 "--clips-per-gpu 64 --steps 1" X=1
run beat "--config beat --steps 1" X=1
run twh "--config twh --steps 1" X=1
