#!/usr/bin/env python3
"""Is a row's result independent of the batch it rides in, within the STREAM set?  (diagnostic for tests/test_gpu_round4.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
for cfgname, Bs in (("tiny", (100, 170, 200, 480)), ("zeggs", (32, 48))):
    cfg = C.CONFIGS[cfgname]
    sd = synth_state_dict(cfg, 20240)
    small = DSGDenoiser(cfg, precision="bf16", max_batch=4, device=0).set_kernel_set("stream")
    small.load_state_dict(sd)
    for B in Bs:
        yb = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
        xb = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 2 + 3) % 1000
        big = DSGDenoiser(cfg, precision="bf16", max_batch=B, device=0).set_kernel_set("stream")
        big.load_state_dict(sd)
        out = np.asarray(big(xb, ts, yb))
        for lo in (0, B // 2, B - 4):
            ys = {k: (v[lo:lo + 4] if v.shape[0] == B else v) for k, v in yb.items()}
            want = np.asarray(small(xb[lo:lo + 4], ts[lo:lo + 4], ys))
            nd = int((out[lo:lo + 4] != want).sum())
            rel = float(np.linalg.norm(out[lo:lo+4] - want) / np.linalg.norm(want))
            print(f"{cfgname} B={B} rows={B * (cfg.n_poses + 1)} lo={lo}: differing elements {nd} of {want.size}, rel {rel:.2e}", flush=True)
        del big
