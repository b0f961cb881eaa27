#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r04_c}
for c in beat twh; do
  for v in 0 1; do
  DSG_TNW2=$v python tools/aql_timeline.py --config $c --kset tile --steps 300 --first 100 --n 16 --out $O/${T}_timeline_${c}_tile_tnw$v.json > $O/${T}_timeline_${c}_tile_tnw$v.log 2>&1
  f=${c}_tile_tnw$v; echo "== $f"; grep -E "^ *[0-9]+ " $O/${T}_timeline_$f.log | head -8; grep -E "^ *[0-9]+ " $O/${T}_timeline_$f.log | tail -1; grep -A3 -E "us_per_step_untraced" $O/${T}_timeline_$f.log | tr -d '\n'; echo; grep -E "sum_busy_us|per_boundary|packets_per_step|kernel_set|identical" $O/${T}_timeline_$f.log | tr -d '\n'; echo
  done
done
