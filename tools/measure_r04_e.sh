#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r04_e}
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_wavlm.py tests/test_gpu_round3.py -m gpu -q -x -s -p no:cacheprovider --durations=8 > $O/${T}_pytest_gpu_new.log 2>&1
tail -25 $O/${T}_pytest_gpu_new.log
timeout 600 python tools/e2e.py --reps 3 > $O/${T}_e2e_fp32wavlm.log 2>&1; tail -1 $O/${T}_e2e_fp32wavlm.log
timeout 600 python tools/e2e.py --reps 3 --wavlm-dtype bf16 > $O/${T}_e2e_bf16wavlm.log 2>&1; tail -1 $O/${T}_e2e_bf16wavlm.log
