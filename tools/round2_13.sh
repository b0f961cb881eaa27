#!/bin/bash
TAG=${1:-r02_o}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
./tools/xcc_probe > $O/${TAG}_xcc_probe.log 2>&1; cut -c1-300 $O/${TAG}_xcc_probe.log
DSG_PIN_NOCHECK=1 timeout 600 python tools/pin_check.py --lanes 8,16,32 --windows 1 > $O/${TAG}_pin_nocheck.log 2>&1
grep -v "^$" $O/${TAG}_pin_nocheck.log | tail -12
rm -rf $O/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o z -- python tools/pin_check.py --lanes 1 --windows 1 > $O/${TAG}_prof.log 2>&1
find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_pin1_kernel_stats.csv \;
find $O -name "*_kernel_trace.csv" -delete 2>/dev/null
head -16 $O/${TAG}_pin1_kernel_stats.csv | cut -c1-150
