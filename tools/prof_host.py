#!/usr/bin/env python3
"""Host time of the window loop outside the step loops: wall time per pass of `generate_clip` / `generate_clips_streams` against the device time
of the step loops inside it (AQL: first doorbell -> completion signal), with a cProfile of the host side.
    python tools/prof_host.py [--lanes 4] [--batch 4] [--sampler ddpm|ddim50] [--skip 900] [--passes 4]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from diffusestylegesture_amd import config as C
from diffusestylegesture_amd import sample as S
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

p = argparse.ArgumentParser()
p.add_argument("--lanes", type=int, default=4)
p.add_argument("--batch", type=int, default=4)
p.add_argument("--sampler", default="ddpm")
p.add_argument("--skip", type=int, default=900)
p.add_argument("--passes", type=int, default=4)
p.add_argument("--top", type=int, default=18)
a = p.parse_args()
cfg = C.ZEGGS
m = DSGDenoiser(cfg, precision="bf16", max_batch=a.batch, device=0)
m.load_state_dict(synth_state_dict(cfg, 20240))
lanes = [m] + [m.clone() for _ in range(a.lanes - 1)]
ddim = a.sampler == "ddim50"
d = create_gaussian_diffusion(timestep_respacing="ddim50" if ddim else "")
skip = 0 if ddim else a.skip
feats = [[torch.from_numpy(synth_window_inputs(cfg, a.batch, window=w, clip0=ln * a.batch)["audio"]).cuda() for w in range(4)] for ln in range(a.lanes)]
style = torch.tensor([[1.0] + [0.0] * (cfg.style_dim_in - 1)] * a.batch).cuda()
loop_ms = []


def one(i):
    if a.lanes > 1:
        out = S.generate_clips_streams(lanes, d, feats, style, seed=100 + i, smoothing=True, skip_timesteps=skip, stream_ids=list(range(a.lanes)), ddim=ddim)
    else:
        out = S.generate_clip(m, d, feats[0], style, seed=100 + i, smoothing=True, skip_timesteps=skip, sample_fn=d.ddim_sample_loop if ddim else d.p_sample_loop)
    return out


for i in range(2):
    one(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(a.passes):
    one(2 + i)
torch.cuda.synchronize()
pr.disable()
wall = 1e3 * (time.perf_counter() - t0) / a.passes
ms, n = max(ln.last_sample_ms() for ln in lanes)
print(f"{a.lanes} lanes x batch {a.batch} {a.sampler} skip {skip}: {wall:.3f} ms per pass of 4 windows; step loop of the last window {ms:.3f} ms ({n} steps) "
      f"-> host outside the loops ~ {wall - 4 * ms:.3f} ms per pass")
pstats.Stats(pr).sort_stats("tottime").print_stats(a.top)
