#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 0; do
echo "== DSG_FFN=$v"
DSG_FFN=$v python tools/aql_timeline.py --batch 16 --n 16 --out $O/r04_m_timeline_b16_ffn$v.json 2>&1 | grep -E "^ *[0-9]+ " | sed -n 1,9p
DSG_FFN=$v timeout 600 python tools/sweep.py --steps 150 --reps 3 --spec block:1x16,block:4x4,block:4x8,block:1x32,block:4x16 2>&1 | grep -v amdgpu.ids | cut -c1-125
DSG_FFN=$v timeout 600 python tools/sweep.py --steps 50 --reps 3 --sampler ddim50 --spec block:1x16 2>&1 | grep -v amdgpu.ids | cut -c1-125
done
python -m pytest tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "config3 or kernel_sets or guidance" 2>&1 | tail -2
