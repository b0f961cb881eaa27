#!/bin/bash
# timing probe: what would same-XCD, L2-resident hand-offs buy the STREAM step?  (values are NOT valid in the probe modes)
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "1 0" "3 0" "3 1"; do set -- $v
  echo "== DSG_UC=$1 DSG_WS2_REMAP=$2"
  DSG_UC=$1 DSG_WS2_REMAP=$2 python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/r04_i_timeline_b64_uc$1_remap$2.json 2>&1 | grep -E "^ *[0-9]+ |us_per_step_untraced" | sed -n 1,14p
  DSG_UC=$1 DSG_WS2_REMAP=$2 python tools/aql_timeline.py --batch 16 --kset stream --steps 120 --first 40 --n 16 --out $O/r04_i_timeline_b16_uc$1_remap$2.json 2>&1 | grep -E "us_per_step_untraced" -A1 | tr -d '\n'; echo
done
