#!/bin/bash
TAG=${1:-r02_p}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/pin_check.py --lanes 8 --windows 1 > $O/${TAG}_pin8.log 2>&1; grep -v "^$" $O/${TAG}_pin8.log | tail -5
DSG_PIN_FENCED=1 DSG_PIN_NOCHECK=1 timeout 300 python tools/pin_check.py --lanes 1,8 --windows 1 > $O/${TAG}_pin_fenced.log 2>&1; grep -v "^$" $O/${TAG}_pin_fenced.log | tail -6
