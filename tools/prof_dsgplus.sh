cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in beat twh; do
rm -rf gpurun_out/prof_$c
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -o z -- python tools/step_timing.py --config $c --steps 100 --reps 1 > gpurun_out/prof_$c.log 2>&1
echo "== $c"; find gpurun_out/prof_$c -name "*kernel_stats.csv" -exec head -9 {} \; | cut -c1-140
find gpurun_out/prof_$c -name "*_kernel_trace.csv" -delete
done
