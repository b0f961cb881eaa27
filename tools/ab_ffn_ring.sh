#!/bin/bash
# A/B of k_ffn<OP> on one box: weights on double-buffered groups (DSG_FFN_RING=0) / on one rolling ring of fragments (=1) -- in-kernel stamps at 64 clips
# (32-row form), bench lines at 1 x 64, 4 x 16, 4 x 32 (32-row form) and 4 x 48, 4 x 64 clips (64-row form).    gpurun --timeout 600 -- 'bash tools/ab_ffn_ring.sh r05_s'
TAG=${1:-r05_s}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python - $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["us_per_step_untraced_runs"], {k: (v["launches_per_step"], round(v["busy_us"] / v["launches_per_step"], 2)) for k, v in d["by_kernel"].items()})
PY
}
B="timeout 200 python bench.py --no-cpu-baseline --sub-records off --no-postprocess"
line() { grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1:', j['value'], j['us_per_denoise_step'])"; }
if [ "${2:-}" = "big" ]; then
  for E in 0 1 0 1; do
    DSG_FFN_RING=$E $B --clips-per-gpu 256 --steps 1 --warmup 1 2>&1 | line "4x64 ring$E"
  done
  for E in 0 1; do
    DSG_FFN_RING=$E $B --clips-per-gpu 192 --steps 1 --warmup 1 2>&1 | line "4x48 ring$E"
    DSG_FFN_RING=$E DSG_FFN_RT4=1 $B --clips-per-gpu 64 --lanes 1 --steps 1 --warmup 1 2>&1 | line "1x64 64-row ring$E"
  done
else
  for E in 0 1; do
    DSG_FFN_RING=$E python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/${TAG}_timeline_b64_stream_ring$E.json > /dev/null 2>&1; show $O/${TAG}_timeline_b64_stream_ring$E.json
  done
  for E in 0 1; do
    DSG_FFN_RING=$E $B --clips-per-gpu 64 --lanes 1 --steps 1 --warmup 1 2>&1 | line "1x64 ring$E"
    DSG_FFN_RING=$E $B --clips-per-gpu 64 --steps 1 --warmup 1 2>&1 | line "4x16 ring$E"
    DSG_FFN_RING=$E $B --clips-per-gpu 128 --steps 1 --warmup 1 2>&1 | line "4x32 ring$E"
    DSG_FFN_RING=$E DSG_FFN_RT4=0 $B --clips-per-gpu 256 --steps 1 --warmup 1 2>&1 | line "4x64 32-row ring$E"
  done
fi
timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider -k "ffn_64_row" 2>&1 | tail -3
