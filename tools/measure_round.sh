#!/bin/bash
# One measurement round on the GPU box; everything lands in gpurun_out/$TAG_* (copy what matters into profiles/).
#   gpurun --timeout 1200 -- 'bash tools/measure_round.sh r01_h'
TAG=${1:-rXX}
O=gpurun_out
mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > $O/${TAG}_bench.log 2>&1
python bench.py --sampler ddim50 --batch 16 --no-cpu-baseline > $O/${TAG}_bench_ddim50_b16.log 2>&1
python bench.py --config beat --steps 1 --no-cpu-baseline > $O/${TAG}_bench_beat.log 2>&1
python bench.py --config twh --steps 1 --no-cpu-baseline > $O/${TAG}_bench_twh.log 2>&1
python tools/step_timing.py --latency on,off --reps 3 > $O/${TAG}_step_timing.log 2>&1
DSG_AQL=0 python tools/step_timing.py --latency on --reps 3 > $O/${TAG}_step_timing_hip_launches.log 2>&1
timeout 100 tools/_build/aql_probe tools/_build/aql_kernels.hsaco > $O/${TAG}_aql_probe.log 2>&1
python tools/kernel_chain.py > $O/${TAG}_kernel_chain_b1.log 2>&1
python tools/gpu_check.py > $O/${TAG}_parity_matrix.log 2>&1
DSG_LIB=diffusestylegesture_amd/csrc/libdsg_hip_stamps.so python tools/stamps.py > $O/${TAG}_stamps.log 2>&1
timeout 100 tools/_build/dep_probe > $O/${TAG}_dep_probe.log 2>&1
timeout 100 tools/_build/icache_probe > $O/${TAG}_icache_probe.log 2>&1
rm -rf $O/prof_$TAG $O/pmc_f_$TAG $O/pmc_w_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o z -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$TAG -o z -- python tools/step_timing.py --steps 100 --reps 1 --spg=-1 > $O/${TAG}_pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f_$TAG $O/pmc_w_$TAG 100 > $O/${TAG}_traffic_zeggs_b1_bf16.json 2>$O/${TAG}_traffic.err
find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
# the per-dispatch traces are large; keep only the summaries
find $O/prof_$TAG $O/pmc_f_$TAG $O/pmc_w_$TAG -name "*_kernel_trace.csv" -delete 2>/dev/null
find $O/pmc_f_$TAG $O/pmc_w_$TAG -name "*counter_collection.csv" -delete 2>/dev/null
tail -2 $O/${TAG}_bench.log
