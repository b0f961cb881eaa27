"""Host-side mirror of the reference sampler interface, backed by the HIP library.

Reference interface mirrored (same names, argument order, defaults and error behaviour):
  * `get_named_beta_schedule`, `betas_for_alpha_bar`          main/diffusion/gaussian_diffusion.py:21-65
  * `space_timesteps`, `SpacedDiffusion`                       main/diffusion/respace.py:8-114
  * `GaussianDiffusion.p_sample_loop` / `ddim_sample_loop`     main/diffusion/gaussian_diffusion.py:608-671, :889-936
  * `q_sample`, `_predict_xstart_from_eps`, `q_posterior_mean_variance`, `p_sample`, `ddim_sample`
                                                               :236-278, :400-405, :506-558, :742-792
  * `create_gaussian_diffusion()`                              main/utils/model_util.py:59-100
The whole step loop of `p_sample_loop(model=DSGDenoiser, ...)` runs inside libdsg_hip.so (one hipGraph replay per
`steps_per_graph` steps, no host involvement per step).  Handing any other callable as `model` runs the generic
loop: the callable produces x0 and the fused HIP elementwise kernels (dsg_posterior_step / dsg_ddim_step /
dsg_q_sample) do the sampler arithmetic on the device tensors.

Noise: the reference consumes torch's global generator.  Here every draw comes from the framework's counter-based
stream (Philox4x32-10, see csrc/dsg_kernels.h) addressed by (seed, stream_id, draw index).  `manual_seed(seed)`
plays the role of `torch.manual_seed(seed)` (sample.py:212): it resets the draw counter, and each sampling call
advances it by 1 + n_steps, so consecutive windows of a clip continue one stream exactly like the reference does.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import lib as L


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)],
                    dtype=np.float64)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """Kept-step set; "ddimN" = the DDIM paper's fixed stride, "a,b,c" = per-section counts."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    steps, start = [], 0
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            steps.append(start + round(pos))
            pos += stride
        start += size
    return set(steps)


_TABLES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
           "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
           "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


class DSGDiffusion:
    """SpacedDiffusion(use_timesteps, betas=...) for START_X / FIXED_SMALL models (the only configuration
    `create_gaussian_diffusion` builds).  Table attributes carry the reference's names."""

    def __init__(self, use_timesteps, betas, library: L.DSGLibrary | None = None):
        base = np.array(betas, dtype=np.float64)
        if base.ndim != 1:
            raise ValueError("betas must be 1-D")
        if not ((base > 0).all() and (base <= 1).all()):
            raise ValueError("betas must be in (0, 1]")
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(base)
        ac = np.cumprod(1.0 - base)
        last, nb, tmap = 1.0, [], []
        for i, a in enumerate(ac):
            if i in self.use_timesteps:
                nb.append(1 - a / last)
                last = a
                tmap.append(i)
        self.timestep_map = tmap
        self.num_timesteps = len(nb)
        self._lib = library
        self._set_tables(np.array(nb, dtype=np.float64))
        self.rescale_timesteps = False
        self._seed, self._draw, self.stream_id = 0, 0, 0
        self.last_sample_ms = None

    # the tables are computed by the library's own host code (dsg_schedule_tables), the same code dsg_set_schedule
    # uses on the device path -- so the CPU tests pin exactly what the sampler consumes
    def _set_tables(self, betas):
        n = len(betas)
        lib = self._lib or L.default_library()
        self._lib = lib
        out = np.zeros((11, n), dtype=np.float64)
        lib.check(lib.cdll.dsg_schedule_tables(betas.ctypes.data, n, out.ctypes.data))
        for i, k in enumerate(_TABLES):
            setattr(self, k, out[i].copy())

    # ---- RNG stream ------------------------------------------------------------------------------------------
    def manual_seed(self, seed: int, stream_id: int = 0):
        self._seed, self._draw, self.stream_id = int(seed), 0, int(stream_id)
        return self

    # ---- fused loops -----------------------------------------------------------------------------------------
    def _check_unsupported(self, clip_denoised, denoised_fn, cond_fn, randomize_class, cond_fn_with_grad):
        if clip_denoised:
            raise NotImplementedError("clip_denoised=True is not on the sampling path (sample.py:256 passes False)")
        if denoised_fn is not None or cond_fn is not None or randomize_class or cond_fn_with_grad:
            raise NotImplementedError("denoised_fn / cond_fn / randomize_class / cond_fn_with_grad are not supported")

    def _fused(self, mode, model, shape, noise, model_kwargs, skip_timesteps, init_image, dump_steps, const_noise,
               eta, step_noise, seed, draw_base):
        B = int(shape[0])
        if tuple(shape) != (B, model.njoints, model.nfeats, model.cfg.n_poses):
            raise ValueError(f"shape {tuple(shape)} does not match the denoiser ({model.njoints}, {model.nfeats}, "
                             f"{model.cfg.n_poses})")
        y = (model_kwargs or {}).get("y")
        if y is None:
            raise ValueError("model_kwargs['y'] is required")
        model.set_schedule(self)
        model.set_cond(y, B)
        n_run = self.num_timesteps - skip_timesteps
        use_torch = any(L.is_torch(v) for v in (noise, init_image, y.get("audio")))
        nb, ib, sb = L.Buf(noise), L.Buf(init_image), L.Buf(step_noise)
        a = L.dsg_sample_args()
        a.mode, a.skip_timesteps, a.eta, a.const_noise = mode, int(skip_timesteps), float(eta), int(bool(const_noise))
        a.init_noise, a.step_noise, a.init_image = nb.ptr, sb.ptr, ib.ptr
        a.seed = (self._seed if seed is None else int(seed)) & (2 ** 64 - 1)
        a.stream_id = self.stream_id
        a.draw_base = self._draw if draw_base is None else int(draw_base)
        dump = None
        if dump_steps is not None:
            ds = np.ascontiguousarray(sorted(int(d) for d in dump_steps), dtype=np.int32)
            dump = np.zeros((len(ds),) + tuple(shape), dtype=np.float32)
            a.n_dump, a.dump_steps, a.dump_out = len(ds), ds.ctypes.data, dump.ctypes.data
        out, out_ptr = model._alloc_out(shape, use_torch)
        lib = model.lib
        lib.check(lib.cdll.dsg_sample(model.handle, C.byref(a), out_ptr, B, L.current_stream_ptr() if use_torch else None))
        if draw_base is None:
            self._draw += 1 + n_run
        self._last_model = model
        if dump_steps is not None:
            lib.check(lib.cdll.dsg_sync(model.handle))
            res = [dump[i] for i in range(len(dump))]
            if use_torch:
                import torch
                res = [torch.from_numpy(d).to(out.device) for d in res]
            return res
        return out

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                      randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                      *, step_noise=None, seed=None, draw_base=None):
        self._check_unsupported(clip_denoised, denoised_fn, cond_fn, randomize_class, cond_fn_with_grad)
        from .model import DSGDenoiser
        if isinstance(model, DSGDenoiser):
            return self._fused(L.MODE_DDPM, model, shape, noise, model_kwargs, skip_timesteps, init_image,
                               dump_steps, const_noise, 0.0, step_noise, seed, draw_base)
        return self._generic_loop(False, model, shape, noise, model_kwargs, skip_timesteps, init_image, dump_steps,
                                  const_noise, 0.0, device)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                         *, step_noise=None, seed=None, draw_base=None):
        if dump_steps is not None:
            raise NotImplementedError()
        if const_noise:
            raise NotImplementedError()
        self._check_unsupported(clip_denoised, denoised_fn, cond_fn, randomize_class, cond_fn_with_grad)
        from .model import DSGDenoiser
        if isinstance(model, DSGDenoiser):
            return self._fused(L.MODE_DDIM, model, shape, noise, model_kwargs, skip_timesteps, init_image, None,
                               False, eta, step_noise, seed, draw_base)
        return self._generic_loop(True, model, shape, noise, model_kwargs, skip_timesteps, init_image, None, False,
                                  eta, device)

    def last_step_time_us(self):
        """GPU time per denoising step of the last fused call (HIP events inside the library)."""
        m = getattr(self, "_last_model", None)
        if m is None:
            return None
        ms, n = C.c_float(), C.c_int()
        m.lib.check(m.lib.cdll.dsg_last_sample_ms(m.handle, C.byref(ms), C.byref(n)))
        return 1000.0 * ms.value / max(n.value, 1)

    # ---- generic loop: any callable model, fused HIP elementwise kernels for the sampler arithmetic --------------
    def _f32(self, name, idx, B):
        return np.full((B,), np.float32(getattr(self, name)[idx]), dtype=np.float32)

    def _generic_loop(self, ddim, model, shape, noise, model_kwargs, skip_timesteps, init_image, dump_steps,
                      const_noise, eta, device):
        import torch
        lib = self._lib or L.default_library()
        if device is None:
            device = next(model.parameters()).device
        B = int(shape[0])
        per = int(np.prod(shape[1:]))
        stream = L.current_stream_ptr()
        z = lambda: torch.randn(*shape, device=device)
        img = noise if noise is not None else z()
        if skip_timesteps and init_image is None:
            init_image = torch.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            out = torch.empty_like(img)
            lib.check(lib.cdll.dsg_q_sample(out.data_ptr(), init_image.contiguous().data_ptr(),
                                            img.contiguous().data_ptr(),
                                            self._f32("sqrt_alphas_cumprod", indices[0], B).ctypes.data,
                                            self._f32("sqrt_one_minus_alphas_cumprod", indices[0], B).ctypes.data,
                                            B, per, stream))
            img = out
        tmap = torch.tensor(self.timestep_map, device=device, dtype=torch.long)
        dump = []
        for n, i in enumerate(indices):
            t = torch.full((B,), i, device=device, dtype=torch.long)
            with torch.no_grad():
                x0 = model(img, tmap[t], **(model_kwargs or {})).contiguous().float()
            eps = z()
            if const_noise:
                eps = eps[[0]].repeat(B, 1, 1, 1)
            nz = np.float32(0.0 if i == 0 else 1.0)
            out = torch.empty_like(img)
            img = img.contiguous()
            if not ddim:
                sig = nz * np.exp(np.float32(0.5) * np.float32(self.posterior_log_variance_clipped[i]))
                lib.check(lib.cdll.dsg_posterior_step(
                    out.data_ptr(), x0.data_ptr(), img.data_ptr(), eps.data_ptr(),
                    self._f32("posterior_mean_coef1", i, B).ctypes.data,
                    self._f32("posterior_mean_coef2", i, B).ctypes.data,
                    np.full((B,), sig, np.float32).ctypes.data, B, per, stream))
            else:
                ab, abp = np.float32(self.alphas_cumprod[i]), np.float32(self.alphas_cumprod_prev[i])
                one = np.float32(1)
                sigma = np.float32(eta) * np.sqrt((one - abp) / (one - ab)) * np.sqrt(one - ab / abp)
                coef = np.tile(np.array([np.float32(self.sqrt_recip_alphas_cumprod[i]),
                                         np.float32(self.sqrt_recipm1_alphas_cumprod[i]), np.sqrt(abp),
                                         np.sqrt(one - abp - sigma * sigma), nz * sigma], np.float32), (B, 1))
                lib.check(lib.cdll.dsg_ddim_step(out.data_ptr(), x0.data_ptr(), img.data_ptr(), eps.data_ptr(),
                                                 np.ascontiguousarray(coef).ctypes.data, B, per, stream))
            img = out
            if dump_steps is not None and n in dump_steps:
                dump.append(img.clone())
        return dump if dump_steps is not None else img


def create_gaussian_diffusion(timestep_respacing="", steps=1000, noise_schedule="cosine", library=None):
    """`create_gaussian_diffusion()` of main/utils/model_util.py:59-100 (cosine, 1000 steps, predict x_start,
    FIXED_SMALL, no respacing); `timestep_respacing="ddim50"` gives the DDIM-50 sampler of BASELINE config 3."""
    betas = get_named_beta_schedule(noise_schedule, steps, 1.)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return DSGDiffusion(space_timesteps(steps, timestep_respacing), betas, library=library)
