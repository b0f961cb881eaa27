#!/usr/bin/env python3
"""Phase breakdown of the step kernels from in-kernel cycle stamps (wave 0 of workgroup 8; `make stamps` build).
   DSG_LIB=diffusestylegesture_amd/csrc/libdsg_hip_stamps.so python tools/stamps.py
Rows: kernel id 0 = k_mid, 1+EPI = GEMMs (1 PARTIAL, 2 QKV, 3 RESID(out_proj/linear2), 4 GELU, 5 OUT).  The stamps of the
LAST launch of each kernel in the step survive.  Columns are deltas between consecutive stamps in ns (s_memtime runs at
100 MHz on gfx950? -> calibrated against the HIP-event step time printed next to it)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

cfg = C.CONFIGS["zeggs"]
sd = synth_state_dict(cfg, 20240)
d = create_gaussian_diffusion()
y = {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, 1, window=0, seed_pose_scale=0.1).items()}
m = DSGDenoiser(cfg, precision="bf16", max_batch=1, device=0, latency_mode="on")
m.load_state_dict(sd)
fn = m.lib.cdll.dsg_debug_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
names = {0: "k_mid / k_attn_mid", 1: "gemm PARTIAL", 2: "gemm QKV", 3: "gemm RESID (linear2)", 4: "gemm GELU", 5: "gemm OUT (head)"}
for rep in range(3):
    d.manual_seed(1, 0).p_sample_loop(m, (1, cfg.njoints, 1, cfg.n_poses), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=800)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    rc = fn(m.handle, buf, 128)
    st = np.array(buf[:], dtype=np.int64).reshape(8, 16)
    print(f"rep {rep}: {d.last_step_time_us():.2f} us/step; raw stamp deltas (ticks)")
    for k in range(6):
        r = st[k]
        nz = [i for i in range(16) if r[i] != 0]
        if not nz:
            continue
        base = r[nz[0]]
        order = sorted(nz, key=lambda i: r[i])
        print(f"  {names[k]:22s} " + " ".join(f"[{i}]+{r[i] - base}" for i in order))
