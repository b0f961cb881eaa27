#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
for cfgname, B in (("zeggs", 8), ("zeggs", 48), ("tiny", 8), ("tiny", 200)):
    cfg = C.CONFIGS[cfgname]
    sd = synth_state_dict(cfg, 20240)
    yb = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    xb = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 2 + 3) % 1000
    outs = {}
    for v in ("0", "1"):
        os.environ["DSG_ATTN_OP2"] = v
        m = DSGDenoiser(cfg, precision="bf16", max_batch=B, device=0).set_kernel_set("stream"); m.load_state_dict(sd)
        outs[v] = np.asarray(m(xb, ts, yb)).copy()
        del m
    d = outs["0"] != outs["1"]
    print(f"{cfgname} B={B}: k_attn_op vs k_attn_op2: differing {int(d.sum())} of {d.size}; clips: {np.nonzero(d.reshape(B, -1).any(1))[0][:10]}", flush=True)
