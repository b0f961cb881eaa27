#!/usr/bin/env python3
"""One parameterised A/B tool for the step loop (replaces the per-experiment shell scripts of rounds 1-2): every spec runs in
THIS process on the same GPU, same weights, same conditioning and noise, and prints one line.

    python tools/sweep.py --spec block:1x16,stream:1x16,block:4x4,stream:4x4 --steps 100 [--reps 3] [--sampler ddpm|ddim50]
                          [--config zeggs] [--precision bf16]

spec = <kernel set>:<lanes>x<batch per lane>[:uc0|uc1|uc2][:hip]      (kernel set: auto / latency / tile / block / stream;
       uc = DSG_UC for the handles of this spec; hip = HIP launches instead of AQL packets -- what rocprofv3 can see)
Per spec: us per denoising step (time in which ALL clips of the spec advance one step; best and median of --reps), the
frames/s that corresponds to for 320-frame clips of 4 x 1000 steps, which path / kernel set really ran, and the rel-L2
distance of the samples to those of the FIRST spec with the same lanes x batch (a cross-check of the kernel sets against
each other; parity against the oracle lives in tests/)."""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

p = argparse.ArgumentParser()
p.add_argument("--spec", required=True)
p.add_argument("--steps", type=int, default=100)
p.add_argument("--reps", type=int, default=3)
p.add_argument("--sampler", default="ddpm")
p.add_argument("--config", default="zeggs")
p.add_argument("--precision", default="bf16")
a = p.parse_args()
cfg = C.CONFIGS[a.config]
sd = synth_state_dict(cfg, 20240)
d = create_gaussian_diffusion("ddim50" if a.sampler == "ddim50" else "")
n_steps = min(a.steps, d.num_timesteps)
skip = d.num_timesteps - n_steps
first = {}
for spec in a.spec.split(","):
    parts = spec.split(":")
    kset, (nl, b) = parts[0], (int(v) for v in parts[1].split("x"))
    for k in ("DSG_UC", "DSG_AQL"):
        os.environ.pop(k, None)
    for o in parts[2:]:
        if o.startswith("uc"):
            os.environ["DSG_UC"] = o[2:]
        elif o == "hip":
            os.environ["DSG_AQL"] = "0"
    m = DSGDenoiser(cfg, precision=a.precision, max_batch=b, device=0).set_kernel_set(kset)
    m.load_state_dict(sd)
    lanes = [m] + [m.clone() for _ in range(nl - 1)]
    shape = (b, cfg.njoints, 1, cfg.n_poses)
    ys = [{"y": {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, b, window=1, clip0=ln * b, seed_pose_scale=0.1).items()}}
          for ln in range(nl)]
    us = []
    for r in range(a.reps + 1):          # the first pass warms up (queues, code objects, weights in L2)
        d.manual_seed(1, 0)
        if nl > 1:
            outs = d.p_sample_loop_multi(lanes, shape, ys, seeds=[1] * nl, stream_ids=list(range(nl)), skip_timesteps=skip, ddim=a.sampler == "ddim50")
        else:
            fn = d.ddim_sample_loop if a.sampler == "ddim50" else d.p_sample_loop
            outs = [fn(m, shape, clip_denoised=False, model_kwargs=ys[0], skip_timesteps=skip)]
        torch.cuda.synchronize()
        if r:
            us.append(max(1000.0 * ln.last_sample_ms()[0] / max(ln.last_sample_ms()[1], 1) for ln in lanes))
    res = np.concatenate([np.asarray(o.cpu()) for o in outs])
    key = (nl, b)
    if key not in first:
        first[key] = res
    dist = float(np.linalg.norm(res.astype(np.float64) - first[key]) / np.linalg.norm(first[key]))
    best, med = min(us), float(np.median(us))
    fps = nl * b * 320 / (4000 * best * 1e-6)
    print(f"{spec:28s} {nl}x{b:<3d} {best:8.2f} us/step (median {med:8.2f})  {fps:9.0f} frames/s-equivalent  set={m.last_kernel_set()} "
          f"path={m.last_sample_path()} fence_free={int(m.last_sample_fence_free())} finite={bool(np.isfinite(res).all())} dist_to_first={dist:.2e} sha1={hashlib.sha1(res.tobytes()).hexdigest()[:12]}", flush=True)
    del lanes, m
