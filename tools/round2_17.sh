#!/bin/bash
TAG=${1:-r02_s}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fence_free or pinned or batch16 or zeggs or lanes" > $O/${TAG}_pytest_sel.log 2>&1; tail -3 $O/${TAG}_pytest_sel.log
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"])
PY
}
run c1 "" X=1
run c4 "--clips-per-gpu 4" X=1
run c8_l4 "--clips-per-gpu 8" X=1
run c16_l4 "--clips-per-gpu 16" X=1
run c16_l1 "--clips-per-gpu 16 --lanes 1" X=1
run c32_l4 "--clips-per-gpu 32" X=1
run c32_l4_uc0 "--clips-per-gpu 32" DSG_UC=0
run c64_l4_uc1 "--clips-per-gpu 64" DSG_UC=1
run c128 "--clips-per-gpu 128 --steps 1" X=1
run c128_uc1 "--clips-per-gpu 128 --steps 1" DSG_UC=1
