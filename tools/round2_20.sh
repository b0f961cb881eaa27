#!/bin/bash
TAG=${1:-r02_v}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"], j.get("fence_free_packets"))
PY
}
run c8_l4 "--clips-per-gpu 8" X=1
run c16_l1 "--clips-per-gpu 16 --lanes 1" X=1
run c16_l1_tnw2 "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_TNW=2
run c16_l1_rt4 "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_RT=4
run c16_l1_head "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_MASK=61
run c16_l4 "--clips-per-gpu 16" X=1
run c16_l4_tnw2 "--clips-per-gpu 16" DSG_GEMM_BLK_TNW=2
run c64_l4 "--clips-per-gpu 64 --steps 1" X=1
run c64_l4_tnw2 "--clips-per-gpu 64 --steps 1" DSG_GEMM_BLK_TNW=2
run c64_l4_head "--clips-per-gpu 64 --steps 1" DSG_GEMM_BLK_MASK=61
