#!/bin/bash
# Phase-mark tables of the step kernels (marks build: `make marks`, or the bf16-only `make dev DEVFLAGS=-DDSG_STAMPS=2` with LIB=dev; dsg_kernels.h
# DSG_TL_MARK).   gpurun --timeout 900 -- 'bash tools/measure_marks.sh r06_a [marks|dev] [b1 b16 b64 beat twh]'
TAG=${1:-r06_a}; LIB=${2:-marks}; shift 2
WHAT=${@:-b1 b16 b64 beat twh}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T="python tools/aql_timeline.py --lib $LIB"
for w in $WHAT; do
  case $w in
    b1)   $T --out $O/${TAG}_marks_b1.json > $O/${TAG}_marks_b1.log 2>&1;;
    b16)  $T --batch 16 --n 16 --out $O/${TAG}_marks_b16_block.json > $O/${TAG}_marks_b16_block.log 2>&1;;
    b64)  $T --batch 64 --kset stream --steps 120 --first 40 --n 16 --out $O/${TAG}_marks_b64_stream.json > $O/${TAG}_marks_b64_stream.log 2>&1;;
    beat) $T --config beat --steps 300 --first 100 --n 16 --out $O/${TAG}_marks_beat_b1.json > $O/${TAG}_marks_beat_b1.log 2>&1;;
    twh)  $T --config twh --steps 300 --first 100 --n 16 --out $O/${TAG}_marks_twh_b1.json > $O/${TAG}_marks_twh_b1.log 2>&1;;
  esac
  echo "== $w"; grep -h "busy .* us/launch" $O/${TAG}_marks_*${w#b}*.log | tail -12
done
