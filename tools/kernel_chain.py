#!/usr/bin/env python3
"""Times graph-replayed chains of ONE phase kernel (dsg_debug_chain) to separate launch floor / body / weight coldness."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diffusestylegesture_amd import config as CF
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

names = ["null", "oproj same-weights", "oproj cycling layers", "LN+linear1+GELU", "linear2 (K=1024)", "k_attn", "k_loc",
         "k_in (split-K)", "pose head (LN + N=1152)", "LN+QKV", "k_mid", None, "k_inloc"]      # (id 11 retired with k_qkv_attn)
cfg = CF.ZEGGS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = DSGDenoiser(cfg, precision="bf16", max_batch=B, device=0)
m.load_state_dict(synth_state_dict(cfg, 20240))
d = create_gaussian_diffusion()
y = {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, B, window=0).items()}
shape = (B, cfg.njoints, 1, cfg.n_poses)
d.manual_seed(1, 0).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990)
fn = m.lib.cdll.dsg_debug_chain
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
extra = {20: "alternate oproj/linear2 (iso avg of 1,4)", 21: "cycle oproj,linear2,k_in,LN+QKV (avg of 1,4,7,9)",
         22: "cycle the 6 step kernels (avg of 12,9,5,10,4,8)"}
for g in (1, 0):
    for which, nm in [(i, n) for i, n in enumerate(names) if n] + sorted(extra.items()):
        us = C.c_float()
        rc = fn(m.handle, which, 1280, g, B, C.byref(us))
        print(f"B={B} {'graph' if g else 'eager'} {which:2d} {nm:28s}: {us.value:7.2f} us/launch" if rc == 0 else f"{nm}: rc={rc}", flush=True)
