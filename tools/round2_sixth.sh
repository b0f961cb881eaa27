#!/bin/bash
TAG=${1:-r02_f}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"])
PY
}
run pose "--steps 2" DSG_ECARRY=0
run ecarry "--steps 2" X=1
run ecarry_noenoise "--steps 2" DSG_ECARRY=2
run ecarry_acq0 "--steps 2" DSG_OVL_ACQUIRE=0
run pose2 "--steps 2" DSG_ECARRY=0
