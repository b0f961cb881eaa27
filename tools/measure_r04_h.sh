#!/bin/bash
# A/B: linear2 on 32 x 64 blocks (DSG_BLK_K_CT4), pose head on 16 x 128 tiles (DSG_LEAN_TNW2), at 16 clips
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  echo "== DSG_BLK_K_CT4=$1 DSG_LEAN_TNW2=$2"
  DSG_BLK_K_CT4=$1 DSG_LEAN_TNW2=$2 timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,block:4x4,block:4x8,tile:1x4,block:1x32 2>&1 | grep -v amdgpu.ids
  DSG_BLK_K_CT4=$1 DSG_LEAN_TNW2=$2 timeout 600 python tools/sweep.py --steps 50 --reps 3 --sampler ddim50 --spec block:1x16 2>&1 | grep -v amdgpu.ids
done
