import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--clips-per-gpu", "16", "--lanes", "1", "--sampler", "ddim50", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-postprocess"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:6000])
