#!/bin/bash
TAG=${1:-r02_l}
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
run l1_b16 "--clips-per-gpu 16 --lanes 1" X=1
run l1_b16_noop "--clips-per-gpu 16 --lanes 1" DSG_ATTN_OP=0
run l4_b4 "--clips-per-gpu 16 --lanes 4" X=1
run l4_b4_noop "--clips-per-gpu 16 --lanes 4" DSG_ATTN_OP=0
run l4_b16 "--clips-per-gpu 64 --lanes 4" X=1
run l4_b16_noop "--clips-per-gpu 64 --lanes 4" DSG_ATTN_OP=0
run l1_b8 "--clips-per-gpu 8 --lanes 1" X=1
run l1_b8_noop "--clips-per-gpu 8 --lanes 1" DSG_ATTN_OP=0
run l1_b4 "--clips-per-gpu 4 --lanes 1" X=1
run l1_b4_noop "--clips-per-gpu 4 --lanes 1" DSG_ATTN_OP=0
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "batch16 or ddim50_batch16 or throughput_kernel or lanes" > $O/${TAG}_pytest_gpu.log 2>&1
tail -3 $O/${TAG}_pytest_gpu.log
bash tools/prof_batch.sh 16 $TAG
