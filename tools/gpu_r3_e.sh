#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
for mask in 3 7; do
  echo "== DSG_STREAM_MASK=$mask"
  DSG_STREAM_MASK=$mask timeout 600 python tools/sweep.py --steps 100 --reps 3 --spec block:1x64,stream:1x64,block:1x32,stream:1x32,block:4x16,stream:4x16 2>&1 | grep -v amdgpu.ids | tee $O/sweep_mask$mask.log
done
DSG_STREAM_MASK=7 bash tools/prof.sh r3e_stream7_1x64 stream:1x64:hip 30
