"""GPU debug: intermediates of the generic sampling loop vs the fused loop (tiny dims)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diffusestylegesture_amd import config as C, lib as L
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from oracle import philox
lib = L.default_library()
cfg = C.TINY
B, J, T = 2, cfg.njoints, cfg.n_poses
t = torch.full((B, J, 1, T), 7.0, device="cuda")
rc = lib.cdll.dsg_noise(t.data_ptr(), B, J, T, 77, 3, 5, None)
torch.cuda.synchronize()
ref = philox.normal_bj1t((B, J, 1, T), 77, 5, 3)
print("dsg_noise device rc", rc, "max diff vs oracle", float(np.abs(t.cpu().numpy() - ref).max()))
h = np.zeros((B, J, 1, T), np.float32)
rc = lib.cdll.dsg_noise(h.ctypes.data, B, J, T, 77, 3, 5, None)
print("dsg_noise host rc", rc, float(np.abs(h - ref).max()))
m = DSGDenoiser(cfg, precision="fp32", max_batch=2, device=0)
m.load_state_dict(synth_state_dict(cfg, 20240))
y = synth_window_inputs(cfg, 2, window=0, seed_pose_scale=0.4)
yt = {k: torch.from_numpy(v).cuda() for k, v in y.items()}
shape = (B, J, 1, T)
d = create_gaussian_diffusion()
class W:
    def __call__(self, xx, tt, y=None):
        o = m(xx, tt, y)
        print("   model in", float(xx.abs().mean()), "t", tt.tolist(), "out", float(o.abs().mean()))
        return o
    def parameters(self):
        return m.parameters()
for skip in (997,):
    fused = d.manual_seed(7, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=skip)
    gen = d.manual_seed(7, 2).p_sample_loop(W(), shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=skip)
    print("fused", float(fused.abs().mean()), "gen", float(gen.abs().mean()), "diff", float((fused - gen).abs().max()))
    fn = d.manual_seed(7, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=skip)
    print("fused numpy-in", float(np.abs(fn).mean()), float(np.abs(fn - fused.cpu().numpy()).max()))
