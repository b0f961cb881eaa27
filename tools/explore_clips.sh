O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="timeout 200 python bench.py --no-cpu-baseline --sub-records off --no-postprocess"
$B --clips-per-gpu 256 --lanes 2 --steps 1 --warmup 1 > $O/r05_v_bench_256clips_l2.log 2>&1
$B --clips-per-gpu 320 --lanes 4 --steps 1 --warmup 1 > $O/r05_v_bench_320clips_l4.log 2>&1
$B --clips-per-gpu 384 --lanes 4 --steps 1 --warmup 1 > $O/r05_v_bench_384clips_l4.log 2>&1
$B --clips-per-gpu 512 --lanes 4 --steps 1 --warmup 1 > $O/r05_v_bench_512clips_l4.log 2>&1
$B --clips-per-gpu 384 --lanes 3 --steps 1 --warmup 1 > $O/r05_v_bench_384clips_l3.log 2>&1
for f in $O/r05_v_bench*.log; do echo -n "$f: "; python - $f <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("no json:", open(sys.argv[1]).read()[-400:].replace("\n", " | "))
else:
    j = json.loads(l[-1]); print(j["value"], "frames/s", j["us_per_denoise_step"], "us/step", j["sample_path"], j.get("kernel_set"), j["roofline"]["frac"])
PY
done
