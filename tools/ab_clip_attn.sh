mkdir -p gpurun_out
for v in 0 1 0 1; do
  for spec in "16 4" "16 1" "32 4"; do set -- $spec
    DSG_CLIP_ATTN=$v timeout 200 python bench.py --no-cpu-baseline --sub-records off --no-postprocess --clips-per-gpu $1 --lanes $2 --steps 1 --warmup 1 > gpurun_out/r05_h_clip${v}_$1_$2.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/r05_h_clip${v}_$1_$2.log") if x.startswith("{")]
j=json.loads(l[-1]) if l else None
print("clip_attn=$v clips=$1 lanes=$2", (j["value"], j["us_per_denoise_step"], j["kernel_set"]) if j else open("gpurun_out/r05_h_clip${v}_$1_$2.log").read()[-400:])
PY
  done
done
