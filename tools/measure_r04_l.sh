#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/aql_timeline.py --batch 16 --n 16 --out $O/r04_l_timeline_b16.json 2>&1 | grep -E "^ *[0-9]+ " | head -12
timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,block:4x4,block:4x8,block:1x32 2>&1 | grep -v amdgpu.ids | cut -c1-110
timeout 600 python tools/sweep.py --steps 50 --reps 3 --sampler ddim50 --spec block:1x16 2>&1 | grep -v amdgpu.ids | cut -c1-110
timeout 600 python tools/sweep.py --steps 100 --reps 2 --config beat --spec block:4x4 2>&1 | grep -v amdgpu.ids | cut -c1-110
python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
