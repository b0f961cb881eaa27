#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
for mask in 2 3; do
  echo "== DSG_STREAM_MASK=$mask"
  DSG_STREAM_MASK=$mask timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,stream:1x16,stream:1x16:uc0,block:4x4,stream:4x4,block:2x8,stream:2x8,block:1x8,stream:1x8,block:1x4,stream:1x4,block:1x64,stream:1x64,block:4x16,stream:4x16,block:1x32,stream:1x32,block:4x8,stream:4x8 2>&1 | grep -v amdgpu.ids | tee $O/sweep_mask$mask.log
done
