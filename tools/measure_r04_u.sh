#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_u
for v in 0 1; do
  if [ $v = 1 ]; then export DSG_FFN_NW8=1; else unset DSG_FFN_NW8; fi
  timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec stream:1x64,stream:4x16,stream:4x32,stream:4x64,stream:1x32 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_nw8_$v.log
  python tools/aql_timeline.py --batch 64 --kset stream --steps 120 --first 40 --n 16 2>&1 | grep -E "k_ffn" | head -3
done
export DSG_FFN_NW8=1
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "stream or STREAM" 2>&1 | tail -3
