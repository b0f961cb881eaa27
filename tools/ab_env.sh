#!/bin/bash
# A/B of ONE environment switch of the library on ONE box (the switches are read at dsg_create: INTEGRATION.md), VAR=0 against VAR=1:
#   bench lines for every clips:lanes spec (two rounds, interleaved, so that drift of the box shows), and -- with "timeline:<batch>:<kset>" specs -- the
#   in-kernel timeline of one step (stamps build), per-kernel microseconds per launch.
#   gpurun --timeout 900 -- 'bash tools/ab_env.sh DSG_FFN_RING r05_s 64:1 64:4 128:4 256:4 timeline:64:stream'
#   gpurun --timeout 900 -- 'bash tools/ab_env.sh DSG_CLIP_ATTN r05_k 16:4 16:1 64:1 256:4'
#   gpurun --timeout 900 -- 'bash tools/ab_env.sh DSG_WS_OUT_ONE r05_x 64:1 256:4 timeline:64:stream'
# A switch that needs a companion (the 64-row k_ffn at any size: DSG_FFN_RT4=1) takes it from the caller's environment.
VAR=${1:?environment switch}; TAG=${2:-ab}; shift 2
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="timeout 300 python bench.py --no-cpu-baseline --sub-records off --no-postprocess"
for round in 1 2; do
  for spec in "$@"; do
    case $spec in timeline:*) continue;; esac
    clips=${spec%%:*}; lanes=${spec##*:}
    for v in 0 1; do
      env $VAR=$v $B --clips-per-gpu $clips --lanes $lanes --steps 1 --warmup 1 2>&1 | grep '^{' | tee -a $O/${TAG}_${VAR}_${clips}x${lanes}_$v.log | \
        python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$VAR=$v clips $clips lanes $lanes:', j['value'], 'frames/s', j['us_per_denoise_step'], 'us/step', j.get('kernel_set'))"
    done
  done
done
for spec in "$@"; do
  case $spec in timeline:*) ;; *) continue;; esac
  IFS=: read -r _ batch kset <<< "$spec"
  for v in 0 1; do
    f=$O/${TAG}_${VAR}_timeline_b${batch}_${kset}_$v.json
    env $VAR=$v python tools/aql_timeline.py --batch $batch --kset $kset --steps 120 --first 40 --n 16 --out $f > /dev/null 2>&1
    python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["us_per_step_untraced_runs"], {k: (v["launches_per_step"], round(v["busy_us"] / v["launches_per_step"], 2)) for k, v in d["by_kernel"].items()})
PY
  done
done
