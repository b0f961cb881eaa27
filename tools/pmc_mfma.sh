# MFMA-side counters per kernel (HIP-launch path under the profiler) at batch 1 and 16: how busy are the matrix cores?
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for B in 1 16 64; do
rm -rf gpurun_out/pmc_mfma_b$B
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_b$B -o z -- python tools/step_timing.py --batch $B --steps 20 --reps 1 --kset $( [ $B = 1 ] && echo latency || ( [ $B = 64 ] && echo stream || echo block ) ) > gpurun_out/pmc_mfma_b$B.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_mfma_b$B/*counter_collection.csv")
if not f: print("no counters for B=$B"); print(open("gpurun_out/pmc_mfma_b$B.log").read()[-600:]); raise SystemExit
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:58]; per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
print("B=$B  kernel | launches | MFMA instr/launch | MFMA MOPS(512 flop)/launch | MFMA busy cyc/launch | GUI_ACTIVE(sum 8 XCD)/launch | MFMA busy / (GUI_ACTIVE/8 * 1024 SIMDs) %")
for k, c in per.items():
    if "dsg::k_" not in k or n[k] < 20: continue
    L = n[k]
    gui = c["GRBM_GUI_ACTIVE"] / L
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / L
    print(f"{k:58s} {L:5d} {c['SQ_INSTS_VALU_MFMA_BF16']/L:9.0f} {c['SQ_INSTS_VALU_MFMA_MOPS_BF16']/L:11.0f} {busy:10.0f} {gui:10.0f} {100*busy/max(gui/8*1024,1):7.3f}")
PY
find gpurun_out/pmc_mfma_b$B -name "*.csv" -size +1M -delete
done
