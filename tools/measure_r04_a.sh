#!/bin/bash
# round 4, call A: where the DSG+ batch-1 step goes before any change (timelines of the timed AQL path) + headline on this box
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_a
python tools/aql_timeline.py --config beat --steps 300 --first 100 --n 16 --out $O/${T}_timeline_beat_latency.json > $O/${T}_timeline_beat_latency.log 2>&1
python tools/aql_timeline.py --config beat --kset tile --steps 300 --first 100 --n 16 --out $O/${T}_timeline_beat_tile.json > $O/${T}_timeline_beat_tile.log 2>&1
python tools/aql_timeline.py --config twh --steps 300 --first 100 --n 16 --out $O/${T}_timeline_twh_tile.json > $O/${T}_timeline_twh_tile.log 2>&1
python tools/aql_timeline.py --config twh --kset latency --steps 300 --first 100 --n 16 --out $O/${T}_timeline_twh_latency.json > $O/${T}_timeline_twh_latency.log 2>&1
python tools/aql_timeline.py --steps 300 --first 100 --n 16 --out $O/${T}_timeline_zeggs.json > $O/${T}_timeline_zeggs.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 2 > $O/${T}_bench.log 2>&1
tail -1 $O/${T}_bench.log | cut -c1-400
for f in beat_latency beat_tile twh_tile twh_latency zeggs; do echo "== $f"; grep -E "^ *[0-9]+ " $O/${T}_timeline_$f.log | head -60; grep -E "us_per_step_untraced|sum_busy_us|per_boundary" $O/${T}_timeline_$f.log; done
