#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_r
timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,stream:1x16,block:1x20,stream:1x20,block:1x24,stream:1x24,block:1x32,stream:1x32,block:1x48,stream:1x48,block:4x8,stream:4x8,block:4x12,stream:4x12,block:4x16,stream:4x16,block:4x24,stream:4x24,block:4x32,stream:4x32,tile:4x4,block:4x4,tile:4x2,block:4x2 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_sets.log
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider -k "block or BLOCK or sets or arrangement or rows_do_not or guidance or ffn" > $O/${T}_pytest_block.log 2>&1; tail -3 $O/${T}_pytest_block.log
