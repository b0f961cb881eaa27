#!/bin/bash
# rocprofv3 kernel stats of one sweep.py spec (HIP launches: the profiler cannot see hand-written AQL packets)
#   tools/prof.sh <tag> <spec> [steps [extra sweep.py args, e.g. --config beat]]      -> gpurun_out/<tag>_kernel_stats.csv
export TMPDIR=/tmp
TAG=$1; SPEC=$2; STEPS=${3:-50}; shift 3 2>/dev/null || shift $#
D=/tmp/prof_$TAG; rm -rf $D; mkdir -p gpurun_out
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $OLDPWD/tools/sweep.py --spec $SPEC --steps $STEPS --reps 1 "$@" ) > gpurun_out/${TAG}_prof.log 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
cp "$F" gpurun_out/${TAG}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=re.sub(r'dsg::|void |\(.*\)','',r['Name'])[:70]
    print(f"{n:72s} {int(r['Calls']):6d} {float(r['AverageNs'])/1000:8.2f} us  {float(r['Percentage']):5.1f}%")
PY
