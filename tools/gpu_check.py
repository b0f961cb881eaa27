#!/usr/bin/env python3
"""Quick GPU parity matrix: forward vs goldens for (config, precision, latency mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

def rel(a, b):
    return float(np.linalg.norm(np.asarray(a).astype(np.float64) - b) / np.linalg.norm(b))
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g2 = np.load(os.path.join(G, "g2_forward_zeggs.npz")); g5 = np.load(os.path.join(G, "g5_forward_dsgplus.npz"))
cases = [(C.ZEGGS, 1, [999], 1, 0.5, 4243, g2["b1_t999_out"]), (C.BEAT, 1, [999], 3, 0.1, 32, g5["beat_out"]),
         (C.TWH, 1, [0], 3, 0.1, 32, g5["twh_out"]), (C.TINY4, 2, [500, 500], 3, 0.1, 33, g5["tiny4_out"])]
for cfg, B, ts, win, sps, xs, gold in cases:
    sd = synth_state_dict(cfg, 20240)
    y = synth_window_inputs(cfg, B, window=win, seed_pose_scale=sps)
    x = np.random.RandomState(xs).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    for prec in ("fp32", "bf16"):
        for lm in ("on", "off"):
            m = DSGDenoiser(cfg, precision=prec, max_batch=B, device=0, latency_mode=lm)
            m.load_state_dict(sd)
            o = m(x, np.array(ts), y)
            o2 = m(x, np.array(ts), y)
            print(f"{cfg.name:6s} {prec} latency={lm:3s} rel={rel(o, gold):.3e} finite={bool(np.isfinite(o).all())} "
                  f"repeatable={bool(np.array_equal(o, o2))} nan_count={int(np.isnan(o).sum())}", flush=True)
            del m
