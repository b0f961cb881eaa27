#!/bin/bash
TAG=${1:-r02_m}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"])
PY
}
run l1_b16_ks2 "--clips-per-gpu 16 --lanes 1" X=1
run l1_b16_ks1 "--clips-per-gpu 16 --lanes 1" DSG_KIN_KS=1
run l1_b16_ks5 "--clips-per-gpu 16 --lanes 1" DSG_KIN_KS=5
run l4_b4_ks2 "--clips-per-gpu 16 --lanes 4" X=1
run l4_b4_ks5 "--clips-per-gpu 16 --lanes 4" DSG_KIN_KS=5
run l4_b16_ks2 "--clips-per-gpu 64 --lanes 4" X=1
run l4_b16_ks5 "--clips-per-gpu 64 --lanes 4" DSG_KIN_KS=5
run l4_b16_head "--clips-per-gpu 64 --lanes 4" DSG_GEMM_BLK_MASK=61
run l4_b4_head "--clips-per-gpu 16 --lanes 4" DSG_GEMM_BLK_MASK=61
run l4_b1 "--clips-per-gpu 4 --lanes 4" X=1
run l4_b1_unfused "--clips-per-gpu 4 --lanes 4" DSG_LATENCY_MODE=0 DSG_ATTN_OP=1
run l4_b1_unfused_blk "--clips-per-gpu 4 --lanes 4" DSG_LATENCY_MODE=0 DSG_ATTN_OP=1 DSG_GEMM_BLK=1
run l4_b2 "--clips-per-gpu 8 --lanes 4" X=1
run l4_b2_unfused "--clips-per-gpu 8 --lanes 4" DSG_LATENCY_MODE=0 DSG_ATTN_OP=1
run l4_b2_unfused_blk "--clips-per-gpu 8 --lanes 4" DSG_LATENCY_MODE=0 DSG_ATTN_OP=1 DSG_GEMM_BLK=1
run l4_b3 "--clips-per-gpu 12 --lanes 4" X=1
