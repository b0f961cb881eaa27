#!/usr/bin/env python3
"""HBM traffic per denoising step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, as the
MI355X guide prescribes).  Usage on the GPU box:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w 100 > profiles/r01_traffic.json
Units: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B for wide coalesced
reads (MI355X_MICROARCH.md, HBM section) -> the read side is doubled ("fetch_x2"); both raw and corrected are reported."""
import collections, csv, glob, json, sys

def load(d, name):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"]] += float(r["Counter_Value"])
    return per

fd, wd, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
step_kernels = [k for k in F if any(t in k for t in ("k_gemm", "k_attn", "k_clip_attn", "k_mid", "k_inloc", "k_loc", "k_ws", "k_ln_frag", "k_ffn"))]
fetch = sum(F[k] for k in step_kernels) * 1024 / steps
write = sum(W.get(k, 0.0) for k in step_kernels) * 1024 / steps
out = {"steps": steps, "fetch_bytes_per_step_raw": fetch, "write_bytes_per_step": write,
       "traffic_bytes_per_step_raw": fetch + write, "traffic_bytes_per_step_fetch_x2": 2 * fetch + write,
       "per_kernel_fetch_KiB_per_step": {k[:60]: round(F[k] / steps, 2) for k in step_kernels},
       "per_kernel_write_KiB_per_step": {k[:60]: round(W.get(k, 0.0) / steps, 2) for k in step_kernels}}
print(json.dumps(out, indent=1))
