#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
cfg = C.TINY
sd = synth_state_dict(cfg, 20240)
kset, pre, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
def mk(Bm):
    y = synth_window_inputs(cfg, Bm, window=1, seed_pose_scale=0.3)
    x = np.random.RandomState(5).randn(Bm, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    return x, (np.arange(Bm) * 2 + 3) % 1000, y
small = DSGDenoiser(cfg, precision="bf16", max_batch=4, device=0).set_kernel_set(kset); small.load_state_dict(sd)
if pre:
    xp, tp, yp = mk(pre)
    p = DSGDenoiser(cfg, precision="bf16", max_batch=pre, device=0).set_kernel_set(kset); p.load_state_dict(sd); p(xp, tp, yp); del p
xb, ts, yb = mk(B)
big = DSGDenoiser(cfg, precision="bf16", max_batch=B, device=0).set_kernel_set(kset); big.load_state_dict(sd)
out = np.asarray(big(xb, ts, yb))
bad = []
for lo in range(0, B, 4):
    ys = {k: (v[lo:lo + 4] if v.shape[0] == B else v) for k, v in yb.items()}
    w = np.asarray(small(xb[lo:lo + 4], ts[lo:lo + 4], ys))
    for i in range(4):
        if (out[lo + i] != w[i]).sum() > 100:
            bad.append(lo + i)
print(f"{kset} pre={pre} B={B}: clips wrong: {bad}", flush=True)
