"""`DSGDenoiser`: the MDM denoiser of DiffuseStyleGesture behind the reference's call signature.

Mirrors `MDM.forward(x, timesteps, y=None, uncond_info=False)` (main/model/mdm.py:166) and the DSG+ variant
(`BEAT-TWH-main/model/mdm.py:134`, `y['uncond']`), the `load_state_dict` weight contract
(main/utils/model_util.py:8-12) and the attributes the callers touch (`njoints`, `nfeats`, `parameters()`,
`eval()`, `to()`).  All arithmetic happens in libdsg_hip.so (csrc/); this class only moves pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as L
from .config import DSGConfig


class DSGDenoiser:
    def __init__(self, cfg: DSGConfig, precision: str = "bf16", max_batch: int = 1, device: int = 0,
                 steps_per_graph: int = 0, library: L.DSGLibrary | None = None, latency_mode: str = "auto"):
        self.cfg = cfg
        self.lib = library or L.default_library()
        self.njoints, self.nfeats = cfg.njoints, 1
        self.precision = precision
        self.device_index = device
        c = L.dsg_config()
        c.variant, c.njoints, c.n_poses, c.n_seed = cfg.variant, cfg.njoints, cfg.n_poses, cfg.n_seed
        c.latent_dim, c.audio_src_dim, c.audio_dim = cfg.latent_dim, cfg.audio_src_dim, cfg.audio_dim
        c.style_dim_in, c.window, c.num_layers = cfg.style_dim_in, cfg.window, cfg.num_layers
        c.num_heads, c.ff_size, c.local_heads = cfg.num_heads, cfg.ff_size, cfg.local_heads
        c.pe_max_len, c.train_steps, c.max_batch = cfg.pe_max_len, 1000, max_batch
        c.precision = {"fp32": L.PREC_FP32, "bf16": L.PREC_BF16}[precision]
        c.device, c.steps_per_graph = device, steps_per_graph
        c.latency_mode = {"auto": 0, "off": 1, "on": 2}[latency_mode]
        h = C.c_void_p()
        self.lib.check(self.lib.cdll.dsg_create(C.byref(c), C.byref(h)))
        self.handle = h
        self.max_batch = max_batch
        self._sched_id = None
        self._loaded = set()

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.cdll.dsg_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- nn.Module-ish surface ---------------------------------------------------------------------------------
    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def parameters(self):
        import torch
        dev = torch.device(f"cuda:{self.device_index}") if torch.cuda.is_available() else torch.device("cpu")
        yield torch.empty(0, device=dev)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Feeds every tensor to dsg_load_tensor under its checkpoint key, then repacks (dsg_finalize_weights).
        Unexpected keys raise like `load_model_wo_clip` asserts (model_util.py:11); `clip_model.*` keys are ignored
        the way the reference tolerates them as missing (model_util.py:12)."""
        for name, t in state_dict.items():
            if name.startswith("clip_model."):
                continue
            b = L.Buf(t)
            shape = tuple(int(s) for s in b.obj.shape)
            arr = (C.c_int64 * len(shape))(*shape)
            self.lib.check(self.lib.cdll.dsg_load_tensor(self.handle, name.encode(), b.p, arr, len(shape), 0))
            self._loaded.add(name)
        self.lib.check(self.lib.cdll.dsg_finalize_weights(self.handle))
        return [], []

    def set_schedule(self, diffusion):
        key = (id(diffusion), diffusion.num_timesteps)
        if self._sched_id == key:
            return
        betas = np.ascontiguousarray(diffusion.betas, dtype=np.float64)
        tmap = np.ascontiguousarray(diffusion.timestep_map, dtype=np.int64)
        self.lib.check(self.lib.cdll.dsg_set_schedule(self.handle, betas.ctypes.data, tmap.ctypes.data, len(betas)))
        self._sched_id = key

    def set_cond(self, y: dict, batch: int, uncond: bool = False):
        style, seed, audio = L.Buf(y["style"]), L.Buf(y.get("seed")), L.Buf(y["audio"])
        mask = y.get("mask_local")
        mb = 0
        mbuf = L.Buf(None)
        if mask is not None:
            mbuf = L.Buf(mask, "uint8")
            mb = int(mbuf.obj.shape[0]) if mbuf.obj.ndim == 2 else 1
        exp_audio = (batch, self.cfg.audio_frames, self.cfg.audio_src_dim)
        if tuple(audio.obj.shape) != exp_audio:
            raise ValueError(f"y['audio'] shape {tuple(audio.obj.shape)} != {exp_audio}")
        if tuple(style.obj.shape) != (batch, self.cfg.style_dim_in):
            raise ValueError(f"y['style'] shape {tuple(style.obj.shape)}")
        if self.cfg.n_seed and tuple(seed.obj.shape) != (batch, self.cfg.njoints, 1, self.cfg.n_seed):
            raise ValueError(f"y['seed'] shape {tuple(seed.obj.shape)}")
        uncond = bool(uncond or y.get("uncond", False))
        stream = L.current_stream_ptr() if L.is_torch(y["audio"]) else None
        if self.cfg.variant == 5:            # DiffuseStyleGesture++: y['seed_last'] (BEAT-TWH mdm.py:229)
            if y.get("seed_last") is None:
                raise KeyError("seed_last")  # what the reference's y['seed_last'] lookup raises
            last = L.Buf(y["seed_last"])
            if tuple(last.obj.shape) != (batch, self.cfg.njoints, 1, self.cfg.n_seed):
                raise ValueError(f"y['seed_last'] shape {tuple(last.obj.shape)}")
            self.lib.check(self.lib.cdll.dsg_set_seed_last(self.handle, last.p, batch, stream))
        self.lib.check(self.lib.cdll.dsg_set_window_cond(self.handle, style.p, seed.p, audio.p, mbuf.p, mb, batch,
                                                         int(uncond), stream))

    def _alloc_out(self, shape, use_torch):
        if use_torch:
            import torch
            dev = torch.device(f"cuda:{self.device_index}") if torch.cuda.is_available() else torch.device("cpu")
            out = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
            return out, C.c_void_p(out.data_ptr())
        out = np.empty(tuple(shape), dtype=np.float32)
        return out, C.c_void_p(out.ctypes.data)

    def forward(self, x, timesteps, y=None, uncond_info=False):
        """x [B, njoints, nfeats, n_poses] fp32, timesteps [B] int64, y dict(style, seed, audio, mask_local)
        -> [B, njoints, nfeats, n_poses]"""
        if y is None:
            raise ValueError("y is required")
        use_torch = L.is_torch(x)
        xb, tb = L.Buf(x), L.Buf(timesteps, "int64")
        B = int(xb.obj.shape[0])
        if tuple(xb.obj.shape) != (B, self.njoints, self.nfeats, self.cfg.n_poses):
            raise ValueError(f"x shape {tuple(xb.obj.shape)}")
        assert tuple(tb.obj.shape) == (B,)
        self.set_cond(y, B, uncond=uncond_info)
        out, optr = self._alloc_out(xb.obj.shape, use_torch)
        self.lib.check(self.lib.cdll.dsg_forward(self.handle, xb.p, tb.p, optr, B,
                                                 L.current_stream_ptr() if use_torch else None))
        return out

    __call__ = forward

    def sync(self):
        self.lib.check(self.lib.cdll.dsg_sync(self.handle))


class ClassifierFreeSampleModel:
    """Classifier-free guidance wrapper, sampling only (main/model/cfg_sampler.py:8-31): two evaluations per call,
    `out_uncond + y['scale'] * (out - out_uncond)`, the unconditional one with `y['uncond'] = True` (which zeroes the
    style embedding -- and, for DiffuseStyleGesture, the seed-pose embedding input -- mdm.py:156-164, :180).  The
    reference class asserts `cond_mode in ['text', 'action']` (inherited from MDM) and therefore cannot wrap the gesture
    models at all; this one can.  A wrapped model is an opaque callable to the sampler, so `p_sample_loop` /
    `ddim_sample_loop` take their generic path (HIP elementwise kernels for the update, one library call per evaluation)
    instead of the single fused `dsg_sample` call."""

    def __init__(self, model):
        self.model = model
        self.cfg, self.njoints, self.nfeats = model.cfg, model.njoints, model.nfeats

    def parameters(self):
        return self.model.parameters()

    def forward(self, x, timesteps, y=None):
        if y is None or "scale" not in y:
            raise KeyError("scale")
        y_uncond = dict(y)
        y_uncond["uncond"] = True
        out = self.model(x, timesteps, y)
        out_uncond = self.model(x, timesteps, y_uncond)
        scale = y["scale"]
        scale = scale.view(-1, 1, 1, 1) if L.is_torch(scale) else np.asarray(scale, np.float32).reshape(-1, 1, 1, 1)
        return out_uncond + scale * (out - out_uncond)

    __call__ = forward
