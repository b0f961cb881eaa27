#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_t
timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec stream:1x64,stream:4x16,stream:4x64,block:1x16 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_ddpm.log
timeout 600 python tools/sweep.py --steps 50 --reps 3 --sampler ddim50 --spec stream:1x64,stream:4x16,stream:4x64,block:1x16 2>&1 | grep -v amdgpu.ids | cut -c1-125 | tee $O/${T}_sweep_ddim50_no_noise.log
