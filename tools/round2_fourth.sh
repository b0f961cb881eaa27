#!/bin/bash
# Embedded-space state at batch 1, block-GEMM defaults, full parity suite
TAG=${1:-r02_d}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1
tail -5 $O/${TAG}_pytest_gpu.log
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"])
PY
}
run b1_ecarry "--steps 2" X=1
run b1_pose "--steps 2" DSG_ECARRY=0
run b1_ecarry_hip "--steps 2" DSG_AQL=0
run b1_pose_hip "--steps 2" DSG_AQL=0 DSG_ECARRY=0
run b2_ecarry "--clips-per-gpu 2 --lanes 1" X=1
run b2_pose "--clips-per-gpu 2 --lanes 1" DSG_ECARRY=0
run l4_b1 "--clips-per-gpu 4 --lanes 4" X=1
run b16_default "--clips-per-gpu 16 --lanes 1" X=1
run b16_mask45 "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_MASK=45
run b16_mask39 "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_MASK=39
run b16_mask53 "--clips-per-gpu 16 --lanes 1" DSG_GEMM_BLK_MASK=53
run b8_default "--clips-per-gpu 8 --lanes 1" X=1
run b8_noblk "--clips-per-gpu 8 --lanes 1" DSG_GEMM_BLK=0
run l4_b4 "--clips-per-gpu 16 --lanes 4" X=1
run l4_b16 "--clips-per-gpu 64 --lanes 4" X=1
run twh "--config twh" X=1
run beat "--config beat" X=1
run beat_pose "--config beat" DSG_ECARRY=0
