#!/bin/bash
# XCD-pinned lanes: correctness (bit-identity with the fenced path) and speed
TAG=${1:-r02_n}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/pin_check.py --lanes 1,8,16,32 --windows 2 > $O/${TAG}_pin_check.log 2>&1
grep -v "^$" $O/${TAG}_pin_check.log | tail -30
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"])
PY
}
run c1_pin "" X=1
run c1_nopin "" DSG_PIN=0
run c8_pin "--clips-per-gpu 8 --lanes 8" X=1
run c8_l4b2 "--clips-per-gpu 8 --lanes 4" DSG_PIN=0
run c16_pin "--clips-per-gpu 16 --lanes 16" X=1
run c32_pin "--clips-per-gpu 32 --lanes 32" X=1
run c64_pin "--clips-per-gpu 64 --lanes 64" X=1
