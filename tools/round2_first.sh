#!/bin/bash
# Round-2 measurement round on the GPU box: parity suite, headline bench, config[3] per-GPU share in both variants, lane sweep,
# rocprofv3 kernel stats of the batched step.   gpurun --timeout 1500 -- 'bash tools/round2_first.sh r02_a'
TAG=${1:-r02_x}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/${TAG}_pytest_gpu.log 2>&1
tail -5 $O/${TAG}_pytest_gpu.log
timeout 300 python bench.py > $O/${TAG}_bench.log 2>&1
for n in 2 4 8 16 32; do
  timeout 200 python bench.py --clips-per-gpu $n --mode streams --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_streams$n.log 2>&1
done
timeout 200 python bench.py --clips-per-gpu 16 --mode lockstep --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_lockstep16.log 2>&1
timeout 200 python bench.py --clips-per-gpu 16 --mode lockstep --sampler ddim50 --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_ddim50_lockstep16.log 2>&1
timeout 200 python bench.py --clips-per-gpu 16 --mode streams --sampler ddim50 --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_ddim50_streams16.log 2>&1
DSG_AQL=0 timeout 200 python bench.py --clips-per-gpu 16 --mode streams --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_streams16_hip_launches.log 2>&1
rm -rf $O/prof_b16_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16_$TAG -o z -- python tools/step_timing.py --batch 16 --steps 100 --reps 1 --latency off > $O/${TAG}_prof_b16.log 2>&1
find $O/prof_b16_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_b16_kernel_stats.csv \;
find $O/prof_b16_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
for f in $O/${TAG}_bench*.log; do echo "== $f"; tail -1 $f | cut -c1-400; done
head -14 $O/${TAG}_b16_kernel_stats.csv | cut -c1-160
