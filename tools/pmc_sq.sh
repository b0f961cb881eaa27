# SQ-side counters per kernel (HIP-launch path: profilers disable the AQL path) at batch 1 and batch 16
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for B in 1 16; do
rm -rf gpurun_out/pmc_sq_b$B
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq_b$B -o z -- python tools/step_timing.py --batch $B --steps 20 --reps 1 --kset $( [ $B = 1 ] && echo latency || echo block ) > gpurun_out/pmc_sq_b$B.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq_b$B/*counter_collection.csv")
if not f: print("no counters for B=$B"); raise SystemExit
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:58]; per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
print("B=$B  kernel | launches | waves/launch | GUI_ACTIVE/launch | wave_cycles(quad)/wave | wait_any% | wait_inst% | active% | valu/wave | vmem_rd/wave")
for k, c in per.items():
    if "dsg::k_" not in k or n[k] < 20: continue
    L = n[k]; w = c["SQ_WAVES"]
    wc = c["SQ_WAVE_CYCLES"]
    print(f"{k:58s} {L:5d} {w/L:8.0f} {c['GRBM_GUI_ACTIVE']/L:9.0f} {wc/w:9.0f} {100*c['SQ_WAIT_ANY']/wc:6.1f} {100*c['SQ_WAIT_INST_ANY']/wc:6.1f} {100*c['SQ_ACTIVE_INST_ANY']/wc:6.1f} {c['SQ_INSTS_VALU']/w:8.0f} {c['SQ_INSTS_VMEM_RD']/w:6.0f}")
PY
find gpurun_out/pmc_sq_b$B -name "*.csv" -size +1M -delete
done
