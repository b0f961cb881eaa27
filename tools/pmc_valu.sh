#!/bin/bash
# Instruction mix per kernel (HIP-launch path under rocprofv3): how much of a kernel's time is VALU / MFMA / LDS / VMEM issue?
#   bash tools/pmc_valu.sh TAG BATCH KSET [STEPS]   ->  gpurun_out/TAG_pmc_valu_b<B>_<set>.log
# floor_us(X) = instructions of class X per launch x issue cycles (VALU 4, trans 16, MFMA busy cycles as counted) / (1024 SIMDs x 2.4 GHz): the
# time the class needs if it were spread perfectly over the chip; against the kernel's duration it says which unit bounds the kernel.
TAG=$1; B=$2; KSET=$3; STEPS=${4:-20}
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=$O/pmc_valu_$TAG; rm -rf $D
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $D -o z -- python tools/step_timing.py --batch $B --steps $STEPS --reps 1 --kset $KSET --spg=-1 > $O/${TAG}_pmc_valu.log 2>&1
python - $D $STEPS <<'PY' | tee $O/${TAG}_pmc_valu_b${B}_${KSET}.log
import csv, glob, collections, sys, re
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/*counter_collection.csv")
if not f:
    print("no counters"); raise SystemExit
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f[0])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("dsg::", "").replace("void ", ""))[:52]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1000.0
print(f"{'kernel':52s} {'launch/step':>11s} {'us/launch':>9s} {'VALU/wave':>9s} {'SALU/wave':>9s} {'LDS/wave':>8s} {'VMEM/wave':>9s} {'waves':>7s} {'VALU floor us':>13s} {'active VALU %':>13s}")
for k, c in sorted(per.items(), key=lambda kv: -dur[kv[0]]):
    if not k.startswith("k_") or n[k] < steps: continue
    L = n[k]; w = c["SQ_WAVES"] / L
    valu = c["SQ_INSTS_VALU"] / L
    floor = valu * 4 / (1024 * 2400.0)
    act = 100 * c["SQ_ACTIVE_INST_VALU"] / max(c["SQ_WAVE_CYCLES"], 1)
    print(f"{k:52s} {L / steps:11.1f} {dur[k] / L:9.2f} {valu / w:9.0f} {c['SQ_INSTS_SALU'] / L / w:9.0f} {c['SQ_INSTS_LDS'] / L / w:8.0f} {c['SQ_INSTS_VMEM'] / L / w:9.0f} {w:7.0f} {floor:13.2f} {act:13.1f}")
PY
find $D -name "*.csv" -size +1M -delete
