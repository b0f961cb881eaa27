for B in 3 4 8 16; do for l in 0 1; do echo -n "B=$B LEAN=$l: "; DSG_GEMM_LEAN=$l timeout 120 python tools/step_timing.py --batch $B --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //' | cut -d' ' -f1-2; done; done
DSG_GEMM_LEAN=1 timeout 200 python -m pytest tests -m gpu -q -k "batch16 or throughput or forward_zeggs or dsgplus" 2>&1 | tail -1
