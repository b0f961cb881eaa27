cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b16 -o z -- python tools/step_timing.py --batch 16 --steps 100 --reps 1 --latency off > gpurun_out/prof_b16.log 2>&1
find gpurun_out/prof_b16 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_i_b16_kernel_stats.csv \;
find gpurun_out/prof_b16 -name "*_kernel_trace.csv" -delete
head -14 gpurun_out/r01_i_b16_kernel_stats.csv | cut -c1-150
tail -1 gpurun_out/prof_b16.log
for i in 1 2 3; do timeout 200 python -m pytest tests -m gpu -q 2>&1 | tail -1; done
