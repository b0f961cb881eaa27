#!/bin/bash
TAG=${1:-r02_u}
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
run c1 "" X=1
run c1_nomid "" DSG_FUSE_ATTN_MID=0
run c1_unfused "" DSG_LATENCY_MODE=0
run c8_l4 "--clips-per-gpu 8" X=1
run c8_l4_unfused "--clips-per-gpu 8" DSG_LATENCY_MODE=0
run c8_l4_unfused_op "--clips-per-gpu 8" DSG_LATENCY_MODE=0 DSG_ATTN_OP=1
run c12_l4 "--clips-per-gpu 12" X=1
run c12_l4_fused "--clips-per-gpu 12" DSG_LATENCY_MODE=1
run c16_l4_op0 "--clips-per-gpu 16" DSG_ATTN_OP=0
run c16_l4 "--clips-per-gpu 16" X=1
