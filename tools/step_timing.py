#!/usr/bin/env python3
"""Per-denoising-step timing of one ZEGGS window on the GPU (used under rocprofv3 and for A/B runs).
   python tools/step_timing.py [--precision bf16] [--steps 200] [--spg 20] [--batch 1] [--config zeggs] [--kset auto,tile]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

p = argparse.ArgumentParser()
p.add_argument("--precision", default="bf16")
p.add_argument("--steps", type=int, default=200)
p.add_argument("--spg", default="0", help="comma-separated steps-per-graph values (-1 = eager)")
p.add_argument("--batch", type=int, default=1)
p.add_argument("--config", default="zeggs")
p.add_argument("--reps", type=int, default=3)
p.add_argument("--kset", default="auto", help="comma-separated kernel sets: auto,latency,tile,block,stream")
a = p.parse_args()
cfg = C.CONFIGS[a.config]
sd = synth_state_dict(cfg, 20240)
d = create_gaussian_diffusion()
y = {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, a.batch, window=0, seed_pose_scale=0.1).items()}
shape = (a.batch, cfg.njoints, 1, cfg.n_poses)
print("HIP_FORCE_DEV_KERNARG =", os.environ.get("HIP_FORCE_DEV_KERNARG"), flush=True)
for spg, lm in [(int(v), l) for v in a.spg.split(",") for l in a.kset.split(",")]:
    m = DSGDenoiser(cfg, precision=a.precision, max_batch=a.batch, device=0, steps_per_graph=spg).set_kernel_set(lm)
    m.load_state_dict(sd)
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = d.manual_seed(1, 0).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=1000 - a.steps)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"{a.config} {a.precision} B={a.batch} spg={spg} kset={lm} ran={m.last_kernel_set()} rep{r}: {d.last_step_time_us():.2f} us/step (HIP events), "
              f"wall {1e6 * wall / a.steps:.2f} us/step, finite={bool(torch.isfinite(s).all())}", flush=True)
    del m
