#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_v
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset DSG_FFN_NW8; else export DSG_FFN_NW8=$v; fi
  timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec stream:1x64,stream:4x16,stream:4x64 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_ffn_variant$v.log
  python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 2>&1 | grep -E "^ *[0-9]+ k_ffn" | head -2
done
