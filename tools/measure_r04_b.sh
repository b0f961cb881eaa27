#!/bin/bash
# round 4, call B: DSG+ batch 1 after (1) one fragment batch per K = D GEMM, (2) k_attn_ph + k_gemm_ln4; multi-process lanes
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r04_b}
for c in beat twh; do
  python tools/aql_timeline.py --config $c --steps 300 --first 100 --n 16 --out $O/${T}_timeline_${c}_auto.json > $O/${T}_timeline_${c}_auto.log 2>&1
  python tools/aql_timeline.py --config $c --kset tile --steps 300 --first 100 --n 16 --out $O/${T}_timeline_${c}_tile.json > $O/${T}_timeline_${c}_tile.log 2>&1
done
python tools/aql_timeline.py --steps 300 --first 100 --n 16 --out $O/${T}_timeline_zeggs.json > $O/${T}_timeline_zeggs.log 2>&1
DSG_ATTN_PH=1 python tools/aql_timeline.py --steps 300 --first 100 --n 16 --out $O/${T}_timeline_zeggs_ph.json > $O/${T}_timeline_zeggs_ph.log 2>&1
for f in beat_auto beat_tile twh_auto twh_tile zeggs zeggs_ph; do echo "== $f"; grep -E "^ *[0-9]+ " $O/${T}_timeline_$f.log | head -9; grep -E "^ *[0-9]+ " $O/${T}_timeline_$f.log | tail -2; grep -A3 -E "us_per_step_untraced" $O/${T}_timeline_$f.log | tr -d '\n'; echo; grep -E "sum_busy_us|per_boundary|packets_per_step|kernel_set" $O/${T}_timeline_$f.log | tr -d '\n'; echo; done
B="timeout 300 python bench.py --no-cpu-baseline"
$B --config beat --steps 1 > $O/${T}_bench_beat.log 2>&1
$B --config twh --steps 1 > $O/${T}_bench_twh.log 2>&1
$B --steps 2 > $O/${T}_bench.log 2>&1
for f in bench_beat bench_twh bench; do tail -1 $O/${T}_$f.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$f', j['value'], j['us_per_denoise_step'], j['kernel_set'], j['sample_path'])"; done
# past four queues: processes x lanes x batch (16 clips on the GPU unless noted)
M="timeout 200 python tools/multiproc.py --steps 300 --reps 3"
( $M --procs 1 --lanes 4 --batch 4
  $M --procs 2 --lanes 4 --batch 2
  $M --procs 4 --lanes 4 --batch 1
  $M --procs 2 --lanes 2 --batch 4
  $M --procs 4 --lanes 1 --batch 4
  $M --procs 2 --lanes 4 --batch 2 --cu-mask halves
  $M --procs 2 --lanes 4 --batch 2 --cu-mask interleave
  $M --procs 1 --lanes 4 --batch 1
  $M --procs 2 --lanes 4 --batch 1
  $M --procs 2 --lanes 4 --batch 4
  $M --procs 2 --lanes 4 --batch 8 ) 2>&1 | grep -v amdgpu.ids | tee $O/${T}_multiproc.log
