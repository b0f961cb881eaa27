"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference samplers and clip orchestration.

  * p_sample_loop / p_sample / p_mean_variance (START_X, FIXED_SMALL, clip_denoised=False)
        `main/diffusion/gaussian_diffusion.py:608-740`, `:506-558`, `:280-398`, `:256-278`
  * ddim_sample_loop / ddim_sample                  `:889-1003`, `:742-792`, `:417-421`
  * the sampler hooks: denoised_fn (`:364-370`, before the clamp), cond_fn through condition_mean (`:428-441`, DDPM: mean + variance * grad)
    and condition_score (`:458-480`, DDIM: eps - sqrt(1 - alpha_bar) * grad -> pred_xstart); pinned by tests/golden/g17_sampler_hooks_tiny.npz
  * q_sample (skip_timesteps / init_image start)    `:236-254`, `:706-713`
  * _extract_into_tensor: float64 table -> `.float()` (fp32) at use   `:1607-1620`
  * _WrappedModel timestep mapping                  `main/diffusion/respace.py:117-129`
  * ZEGGS clip orchestration `inference()`          `main/mydiffusion_zeggs/sample.py:210-296`
  * DSG+  clip orchestration `inference()`          `BEAT-TWH-main/mydiffusion_beat_twh/sample.py:44-192`
Arithmetic is fp32 with fp32-rounded coefficients, in the reference's evaluation order.
Noise comes from `noise_fn(draw)` (draw 0 = x_T, draw 1+i = step i), normally oracle.philox.
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package.
"""
from __future__ import annotations

import numpy as np

from .schedule import OracleDiffusion
from . import philox


def _f(a, i):
    return np.float32(a[i])


def p_sample_loop(diff: OracleDiffusion, model, shape, noise_fn, model_kwargs, skip_timesteps=0,
                  init_image=None, noise=None, const_noise=False, dump_steps=None, clip_denoised=False, denoised_fn=None, cond_fn=None):
    t = diff.t
    img = noise_fn(0).astype(np.float32) if noise is None else np.asarray(noise, np.float32)
    if skip_timesteps and init_image is None:
        init_image = np.zeros_like(img)
    indices = list(range(diff.num_timesteps - skip_timesteps))[::-1]
    if init_image is not None:
        i0 = indices[0]
        img = _f(t["sqrt_alphas_cumprod"], i0) * np.asarray(init_image, np.float32) \
            + _f(t["sqrt_one_minus_alphas_cumprod"], i0) * img
    dump = []
    for n, i in enumerate(indices):
        ts = np.full((shape[0],), diff.timestep_map[i], dtype=np.int64)
        x0 = model(img, ts, **model_kwargs).astype(np.float32)
        if denoised_fn is not None:                     # gaussian_diffusion.py:364-366
            x0 = np.asarray(denoised_fn(x0), np.float32)
        if clip_denoised:                               # gaussian_diffusion.py:377-379
            x0 = np.clip(x0, np.float32(-1), np.float32(1))
        mean = _f(t["posterior_mean_coef1"], i) * x0 + _f(t["posterior_mean_coef2"], i) * img
        if cond_fn is not None:                         # condition_mean, gaussian_diffusion.py:428-441 (variance = posterior_variance: FIXED_SMALL)
            mean = mean + _f(t["posterior_variance"], i) * np.asarray(cond_fn(img, ts, **model_kwargs), np.float32)
        eps = noise_fn(1 + n).astype(np.float32)
        if const_noise:
            eps = np.repeat(eps[[0]], shape[0], 0)
        nz = np.float32(0.0 if i == 0 else 1.0)
        sig = np.exp(np.float32(0.5) * _f(t["posterior_log_variance_clipped"], i))
        img = mean + nz * sig * eps
        if dump_steps is not None and n in dump_steps:
            dump.append(img.copy())
    return dump if dump_steps is not None else img


def ddim_sample_loop(diff: OracleDiffusion, model, shape, noise_fn, model_kwargs, eta=0.0,
                     skip_timesteps=0, init_image=None, noise=None, clip_denoised=False, denoised_fn=None, cond_fn=None):
    t = diff.t
    img = noise_fn(0).astype(np.float32) if noise is None else np.asarray(noise, np.float32)
    if skip_timesteps and init_image is None:
        init_image = np.zeros_like(img)
    indices = list(range(diff.num_timesteps - skip_timesteps))[::-1]
    if init_image is not None:
        i0 = indices[0]
        img = _f(t["sqrt_alphas_cumprod"], i0) * np.asarray(init_image, np.float32) \
            + _f(t["sqrt_one_minus_alphas_cumprod"], i0) * img
    eta = np.float32(eta)
    one = np.float32(1.0)
    for n, i in enumerate(indices):
        ts = np.full((shape[0],), diff.timestep_map[i], dtype=np.int64)
        x0 = model(img, ts, **model_kwargs).astype(np.float32)
        if denoised_fn is not None:
            x0 = np.asarray(denoised_fn(x0), np.float32)
        if clip_denoised:
            x0 = np.clip(x0, np.float32(-1), np.float32(1))
        eps = (_f(t["sqrt_recip_alphas_cumprod"], i) * img - x0) / _f(t["sqrt_recipm1_alphas_cumprod"], i)
        if cond_fn is not None:                         # condition_score, gaussian_diffusion.py:458-480
            eps = eps - np.sqrt(one - _f(t["alphas_cumprod"], i)) * np.asarray(cond_fn(img, ts, **model_kwargs), np.float32)
            x0 = _f(t["sqrt_recip_alphas_cumprod"], i) * img - _f(t["sqrt_recipm1_alphas_cumprod"], i) * eps
            eps = (_f(t["sqrt_recip_alphas_cumprod"], i) * img - x0) / _f(t["sqrt_recipm1_alphas_cumprod"], i)
        ab, abp = _f(t["alphas_cumprod"], i), _f(t["alphas_cumprod_prev"], i)
        sigma = eta * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
        z = noise_fn(1 + n).astype(np.float32)
        mean = x0 * np.sqrt(abp) + np.sqrt(one - abp - sigma ** 2) * eps
        nz = np.float32(0.0 if i == 0 else 1.0)
        img = (mean + nz * sigma * z).astype(np.float32)
    return img


def philox_noise_fn(shape, seed, stream=0):
    return lambda d: philox.normal_bj1t(shape, seed, d, stream)


class CFGModel:
    """ClassifierFreeSampleModel.forward (main/model/cfg_sampler.py:19-31): two evaluations, the second with y['uncond'] =
    True, combined as out_uncond + y['scale'].view(-1, 1, 1, 1) * (out - out_uncond)."""

    def __init__(self, model):
        self.model = model

    def __call__(self, x, timesteps, y=None):
        yy = {k: v for k, v in y.items() if k != "scale"}
        out = self.model(x, timesteps, yy)
        out_uncond = self.model(x, timesteps, yy, uncond_info=True)
        sc = np.asarray(y["scale"], np.float32).reshape(-1, 1, 1, 1)
        return out_uncond + sc * (out - out_uncond)


# ------------------------------------------------------------------------------------------------
# clip orchestration
# ------------------------------------------------------------------------------------------------

def zeggs_clip(sample_window, cfg, feats, style, smoothing=True):
    """ZEGGS `inference()` window loop + stitching (sample.py:236-296), minibatch=True, n_seed != 0.

    sample_window(c, y) -> [1, J, 1, T] float32 sample of window c given conditioning y.
    feats: list of K arrays [1, T, A_src] (WavLM features per window, incl. left context).
    returns normalised poses [K*stride - n_seed, J]."""
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    out = []
    for c, feat in enumerate(feats):
        seedp = np.zeros((1, J, 1, S), np.float32) if c == 0 else out[-1][..., -S:].copy()
        y = {"style": np.asarray([style], np.float32), "seed": seedp, "audio": feat,
             "mask_local": np.ones((1, T), bool)}
        s = np.array(sample_window(c, y), np.float32, copy=True)
        if c > 0:
            last = out[-1][..., -S:].copy()
            out[-1] = out[-1][..., :-S]
            if smoothing:
                delta = (s[:, 0:3, :, 0] - last[:, 0:3, :, 0])[..., None]
                s[:, 0:3] = s[:, 0:3] - delta
            n = 1                                   # len(last_poses) == 1: the reference's len() quirk
            for j in range(n):
                s[..., j] = last[..., j] * np.float32((n - j) / (n + 1)) + s[..., j] * np.float32((j + 1) / (n + 1))
        out.append(s)
    out[-1] = out[-1][..., :-S]
    seq = np.vstack(out)                            # [K, J, 1, stride]
    seq = seq.squeeze(2).transpose(0, 2, 1).reshape(1, -1, J)
    return seq[0, S:]


def dsgplus_clip(sample_window, cfg, feats, style, seed0, real_n_frames, seed_last=None):
    """DSG+ `inference()` (BEAT-TWH sample.py:98-192), attention4: no left audio context, no root shift,
    last window kept whole, first S frames dropped, crop to real_n_frames, keep first J/3 features.
    attention5 (DiffuseStyleGesture++): audio[:-S] per window (sample.py:104, :138) + y['seed_last'] (:85-93).
    attention3 (that tree's "DiffuseStyleGesture"): S feature frames of left context -- zeros, then the previous window's
    tail (sample.py:100-102, :132-134)."""
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    out = []
    for c, feat in enumerate(feats):
        seedp = np.asarray(seed0, np.float32) if c == 0 else out[-1][..., -S:].copy()
        y = {"style": np.asarray([style], np.float32), "seed": seedp, "audio": feat,
             "mask_local": np.ones((1, T), bool)}
        if cfg.variant == 3:
            left = np.zeros_like(np.asarray(feat)[:, :S]) if c == 0 else np.asarray(feats[c - 1])[:, -S:]
            y["audio"] = np.ascontiguousarray(np.concatenate((left, np.asarray(feat)), 1))
        if cfg.variant == 5:
            y["audio"] = np.ascontiguousarray(np.asarray(feat)[:, :-S])
            y["seed_last"] = np.asarray(seed_last, np.float32)
        s = np.array(sample_window(c, y), np.float32, copy=True)
        if c > 0:
            last = out[-1][..., -S:].copy()
            out[-1] = out[-1][..., :-S]
            s[..., 0] = last[..., 0] * np.float32(0.5) + s[..., 0] * np.float32(0.5)
        out.append(s)
    seq = np.concatenate([o.squeeze(2).transpose(0, 2, 1)[0] for o in out], 0)   # [(K-1)*stride + T, J]
    seq = seq[S:][:real_n_frames]
    return seq[:, : J // 3]
