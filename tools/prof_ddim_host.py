#!/usr/bin/env python3
"""Where the host time of a short-loop pass goes (DDIM-50, 16 clips in one batch): wall time of every C-ABI call of the
window loop next to the AQL-timed step loop inside it.
   python tools/prof_ddim_host.py [--batch 16] [--passes 6] [--sampler ddim50|ddpm]"""
import argparse
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd import sample as S
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=16)
p.add_argument("--passes", type=int, default=6)
p.add_argument("--sampler", default="ddim50")
a = p.parse_args()
cfg = C.ZEGGS
m = DSGDenoiser(cfg, precision="bf16", max_batch=a.batch, device=0)
m.load_state_dict(synth_state_dict(cfg, 20240))
d = create_gaussian_diffusion(timestep_respacing="ddim50" if a.sampler == "ddim50" else "")
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
cdll = m.lib.cdll
for name in ("dsg_set_window_cond", "dsg_set_seed_last", "dsg_sample", "dsg_sync"):
    if not hasattr(cdll, name):
        continue
    f = getattr(cdll, name)

    def wrap(f=f, name=name):
        def g(*args):
            t0 = time.perf_counter()
            r = f(*args)
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1
            return r
        return g
    setattr(cdll, name, wrap())
def wrap_py(obj, name):
    f = getattr(obj, name)

    def g(*a_, **k_):
        t0 = time.perf_counter()
        r = f(*a_, **k_)
        acc["py:" + name] += time.perf_counter() - t0
        cnt["py:" + name] += 1
        return r
    setattr(obj, name, g)


for obj, name in ((S, "_zeggs_window_y"), (S, "_zeggs_stitch"), (S, "_zeggs_finish"), (d, "_prepare"), (m, "_alloc_out"), (m, "set_cond"), (m, "set_schedule")):
    if hasattr(obj, name):
        wrap_py(obj, name)
feats = [torch.from_numpy(synth_window_inputs(cfg, a.batch, window=w)["audio"]).cuda() for w in range(4)]
style = [1] + [0] * (cfg.style_dim_in - 1)
sample_fn = d.ddim_sample_loop if a.sampler == "ddim50" else d.p_sample_loop
n_steps = 50 if a.sampler == "ddim50" else 1000
for it in range(a.passes + 1):
    if it == 1:
        acc.clear(); cnt.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
    out = S.generate_clip(m, d, feats, style, seed=123456 + it, smoothing=True, sample_fn=sample_fn)
    torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{a.sampler} batch {a.batch}: {1e3 * wall / a.passes:.2f} ms per pass of 4 windows")
for k in sorted(acc, key=lambda k: -acc[k]):
    print(f"  {k:24s} {1e3 * acc[k] / a.passes:8.3f} ms/pass   {cnt[k] / a.passes:5.1f} calls/pass   {1e6 * acc[k] / max(cnt[k], 1):8.1f} us/call")
c_abi = sum(v for k, v in acc.items() if not k.startswith("py:"))
print(f"  python + torch outside the C-ABI: {1e3 * (wall - c_abi) / a.passes:.3f} ms/pass  (py:* rows are inclusive of nested C-ABI calls)")
print(f"  last window: AQL loop {d.last_step_time_us() * n_steps / 1e3:.3f} ms ({d.last_step_time_us():.2f} us/step x {n_steps})")
