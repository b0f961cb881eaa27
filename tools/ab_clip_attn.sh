# A/B of the round-5 attention kernels (k_clip_attn + the out_proj / LayerNorm1 prologues) against the round-4 pair (DSG_CLIP_ATTN=0)
#   gpurun -- 'bash tools/ab_clip_attn.sh 16:4 64:1 ...'      (clips:lanes)
mkdir -p gpurun_out
SPECS="${@:-16:4 16:1 32:4}"
for v in 0 1 0 1; do
  for spec in $SPECS; do c=${spec%%:*}; l=${spec##*:}
    DSG_CLIP_ATTN=$v timeout 300 python bench.py --no-cpu-baseline --sub-records off --no-postprocess --clips-per-gpu $c --lanes $l --steps 1 --warmup 1 > gpurun_out/r05_ab_clip${v}_${c}_${l}.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/r05_ab_clip${v}_${c}_${l}.log") if x.startswith("{")]
j=json.loads(l[-1]) if l else None
print("clip_attn=$v clips=$c lanes=$l", (j["value"], j["us_per_denoise_step"], j["kernel_set"]) if j else open("gpurun_out/r05_ab_clip${v}_${c}_${l}.log").read()[-400:])
PY
  done
done
