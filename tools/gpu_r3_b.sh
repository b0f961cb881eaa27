#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s -k "stream_kernel_set" 2>&1 | tail -15 > $O/pytest_stream.log
cat $O/pytest_stream.log
timeout 900 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,stream:1x16,stream:1x16:uc0,block:4x4,stream:4x4,block:2x8,stream:2x8,block:1x4,stream:1x4,block:1x64,stream:1x64,block:4x16,stream:4x16,stream:4x16:uc1 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout 300 python tools/sweep.py --sampler ddim50 --steps 50 --reps 4 --spec block:1x16,stream:1x16 2>&1 | grep -v amdgpu.ids | tee $O/sweep_ddim.log
