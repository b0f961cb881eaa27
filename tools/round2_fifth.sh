#!/bin/bash
TAG=${1:-r02_e}
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
for i in 1 2; do
run b1_ecarry_$i "--steps 2" X=1
run b1_pose_$i "--steps 2" DSG_ECARRY=0
done
run b2_ecarry "--clips-per-gpu 2 --lanes 1" X=1
run b2_pose "--clips-per-gpu 2 --lanes 1" DSG_ECARRY=0
run beat "--config beat" X=1
run beat_pose "--config beat" DSG_ECARRY=0
run b1_ecarry_hip "--steps 2" DSG_AQL=0
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "ddpm_chain or ddim_chain or aql_step or full_clip" > $O/${TAG}_pytest_gpu.log 2>&1
tail -3 $O/${TAG}_pytest_gpu.log
