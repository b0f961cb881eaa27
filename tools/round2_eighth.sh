#!/bin/bash
TAG=${1:-r02_i}
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
run l4_b4 "--clips-per-gpu 16 --lanes 4" X=1
run l4_b4_blk45 "--clips-per-gpu 16 --lanes 4" DSG_GEMM_BLK=1
run l4_b4_blk127 "--clips-per-gpu 16 --lanes 4" DSG_GEMM_BLK=1 DSG_GEMM_BLK_MASK=127
run l4_b4_blk5 "--clips-per-gpu 16 --lanes 4" DSG_GEMM_BLK=1 DSG_GEMM_BLK_MASK=5
run l4_b8 "--clips-per-gpu 32 --lanes 4" X=1
run l4_b8_blk45 "--clips-per-gpu 32 --lanes 4" DSG_GEMM_BLK=1
run l4_b8_blk127 "--clips-per-gpu 32 --lanes 4" DSG_GEMM_BLK=1 DSG_GEMM_BLK_MASK=127
run l4_b16 "--clips-per-gpu 64 --lanes 4" X=1
run l4_b16_noblk "--clips-per-gpu 64 --lanes 4" DSG_GEMM_BLK=0
run l4_b16_blk127 "--clips-per-gpu 64 --lanes 4" DSG_GEMM_BLK_MASK=127
run l4_b16_blk127_rt4 "--clips-per-gpu 64 --lanes 4" DSG_GEMM_BLK_MASK=127 DSG_GEMM_BLK_RT=4
run l4_b16_rt4 "--clips-per-gpu 64 --lanes 4" DSG_GEMM_BLK_RT=4
run l4_b32 "--clips-per-gpu 128 --lanes 4" X=1
run l4_b32_rt4 "--clips-per-gpu 128 --lanes 4" DSG_GEMM_BLK_RT=4 DSG_GEMM_BLK_MASK=127
run l2_b32 "--clips-per-gpu 64 --lanes 2" X=1
