#!/bin/bash
# rocprofv3 kernel stats of the batched step at batch $1 (HIP launches: the profiler cannot see the AQL packets)
B=${1:-64}; TAG=${2:-r02_j}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $O/prof_b${B}_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b${B}_$TAG -o z -- python tools/step_timing.py --batch $B --steps 50 --reps 1 --latency off > $O/${TAG}_prof_b$B.log 2>&1
find $O/prof_b${B}_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_b${B}_kernel_stats.csv \;
find $O/prof_b${B}_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
head -14 $O/${TAG}_b${B}_kernel_stats.csv | cut -c1-150
tail -2 $O/${TAG}_prof_b$B.log
