#!/bin/bash
TAG=${1:-r02_q}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "DSG_UC=1" "DSG_UC=2"; do
  timeout 300 python tools/pin_check.py --lanes 1 --windows 2 --env "DSG_PIN=0,$v" > $O/${TAG}_$v.log 2>&1; echo "== $v"; grep -v "^$" $O/${TAG}_$v.log | tail -6
done
run() { name=$1; shift; args=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess $args > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["ms_per_step"], "ms/pass", j["sample_path"])
PY
}
run c1 "" X=1
run c1_uc "" DSG_UC=1
run c16_l1 "--clips-per-gpu 16 --lanes 1" X=1
run c16_l1_uc "--clips-per-gpu 16 --lanes 1" DSG_UC=1
run c16_l4 "--clips-per-gpu 16" X=1
run c16_l4_uc "--clips-per-gpu 16" DSG_UC=1
