# batched path: workgroup width sweep (16-col tiles per wave), DDPM step at batch 16 and DDIM-50 batch-16 bench line
for t in 1 2 4; do echo -n "DSG_GEMM_TNW=$t B=16: "; DSG_GEMM_TNW=$t timeout 120 python tools/step_timing.py --batch 16 --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //'; done
echo -n "default B=16: "; timeout 120 python tools/step_timing.py --batch 16 --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //'
echo -n "default B=4 lat on: "; timeout 120 python tools/step_timing.py --batch 4 --steps 100 --reps 3 --latency on 2>&1 | tail -1 | sed 's/.*rep2: //'
echo -n "default B=4 lat off: "; timeout 120 python tools/step_timing.py --batch 4 --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //'
echo -n "TNW=2 B=4 lat off: "; DSG_GEMM_TNW=2 timeout 120 python tools/step_timing.py --batch 4 --steps 100 --reps 3 --latency off 2>&1 | tail -1 | sed 's/.*rep2: //'
