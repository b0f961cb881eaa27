#!/bin/bash
# Round-2, second measurement round: block GEMMs of the batched path, lanes x batch combinations, generic-loop debug.
TAG=${1:-r02_b}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 python tools/debug_generic.py > $O/${TAG}_debug_generic.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "batch16 or ddim50_batch16 or generic_loop or guidance or throughput_kernel or twh_chain or lanes" > $O/${TAG}_pytest_gpu.log 2>&1
tail -5 $O/${TAG}_pytest_gpu.log
B="timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess"
$B --clips-per-gpu 16 --lanes 1 > $O/${TAG}_l1_b16.log 2>&1
DSG_GEMM_BLK=0 $B --clips-per-gpu 16 --lanes 1 > $O/${TAG}_l1_b16_noblk.log 2>&1
DSG_GEMM_BLK_TNW=1 $B --clips-per-gpu 16 --lanes 1 > $O/${TAG}_l1_b16_tnw1.log 2>&1
DSG_GEMM_BLK_TNW=2 $B --clips-per-gpu 16 --lanes 1 > $O/${TAG}_l1_b16_tnw2.log 2>&1
$B --clips-per-gpu 8 --lanes 1 > $O/${TAG}_l1_b8.log 2>&1
DSG_GEMM_BLK=1 $B --clips-per-gpu 4 --lanes 1 > $O/${TAG}_l1_b4_blk.log 2>&1
$B --clips-per-gpu 4 --lanes 1 > $O/${TAG}_l1_b4.log 2>&1
$B --clips-per-gpu 16 --lanes 2 > $O/${TAG}_l2_b8.log 2>&1
$B --clips-per-gpu 16 --lanes 4 > $O/${TAG}_l4_b4.log 2>&1
DSG_GEMM_BLK=1 $B --clips-per-gpu 16 --lanes 4 > $O/${TAG}_l4_b4_blk.log 2>&1
$B --clips-per-gpu 32 --lanes 2 > $O/${TAG}_l2_b16.log 2>&1
$B --clips-per-gpu 64 --lanes 4 > $O/${TAG}_l4_b16.log 2>&1
$B --clips-per-gpu 8 --lanes 4 > $O/${TAG}_l4_b2.log 2>&1
$B --clips-per-gpu 3 --lanes 3 > $O/${TAG}_l3_b1.log 2>&1
$B --clips-per-gpu 5 --lanes 5 > $O/${TAG}_l5_b1.log 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --clips-per-gpu 16 --lanes 1 --sampler ddim50 > $O/${TAG}_ddim50_l1_b16.log 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --clips-per-gpu 16 --lanes 4 --sampler ddim50 > $O/${TAG}_ddim50_l4_b4.log 2>&1
rm -rf $O/prof_b16_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16_$TAG -o z -- python tools/step_timing.py --batch 16 --steps 100 --reps 1 --latency off > $O/${TAG}_prof_b16.log 2>&1
find $O/prof_b16_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_b16_kernel_stats.csv \;
find $O/prof_b16_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
for f in $O/${TAG}_l*.log $O/${TAG}_ddim*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j["roofline"]["bound"], j["roofline"]["frac"])
PY
done
head -12 $O/${TAG}_b16_kernel_stats.csv | cut -c1-140
cat $O/${TAG}_debug_generic.log | tail -20
