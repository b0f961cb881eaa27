#!/bin/bash
# round 3, GPU call A: the full -m gpu suite after the kernel-set refactor / prune + reference bench numbers of this build
set -x
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b1.log 2>&1
python bench.py --steps 2 --warmup 1 --clips-per-gpu 16 > $O/bench_16_l4b4.log 2>&1
python bench.py --steps 2 --warmup 1 --clips-per-gpu 16 --mode lockstep > $O/bench_16_lockstep.log 2>&1
python bench.py --steps 4 --warmup 1 --sampler ddim50 --clips-per-gpu 16 --mode lockstep > $O/bench_ddim50_b16.log 2>&1
grep -h '"value"' $O/bench_*.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['workload'][:70], d['value'], d['us_per_denoise_step'], d.get('kernel_set'))"
