#!/bin/bash
# Block-GEMM variants in the real batch-16 step (which GEMMs gain, tile shapes), generic-loop tests
TAG=${1:-r02_c}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "generic_loop or batch16" > $O/${TAG}_pytest_gpu.log 2>&1
tail -3 $O/${TAG}_pytest_gpu.log
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-postprocess --clips-per-gpu 16 --lanes 1 > $O/${TAG}_$name.log 2>&1; echo -n "$name: "; python - $O/${TAG}_$name.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-300:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step")
PY
}
run noblk DSG_GEMM_BLK=0
run blk_rt2 DSG_GEMM_BLK_RT=2
run blk_rt4 DSG_GEMM_BLK_RT=4
run blk_rt2_tnw2 DSG_GEMM_BLK_RT=2 DSG_GEMM_BLK_TNW=2
run blk_rt4_tnw2 DSG_GEMM_BLK_RT=4 DSG_GEMM_BLK_TNW=2
for m in 1 2 4 8 16 32; do run only$m DSG_GEMM_BLK_MASK=$m; done
for m in 1 2 4 8 16 32; do run only${m}_rt4 DSG_GEMM_BLK_MASK=$m DSG_GEMM_BLK_RT=4; done
rm -rf $O/prof_b16_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16_$TAG -o z -- python tools/step_timing.py --batch 16 --steps 100 --reps 1 --latency off > $O/${TAG}_prof_b16.log 2>&1
find $O/prof_b16_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_b16_kernel_stats.csv \;
find $O/prof_b16_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
head -12 $O/${TAG}_b16_kernel_stats.csv | cut -c1-140
