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
                 steps_per_graph: int = 0, library: L.DSGLibrary | None = None, latency_mode: str = "auto",
                 _clone_of: "DSGDenoiser | None" = None):
        self.cfg = cfg
        self.lib = library or L.default_library()
        self.njoints, self.nfeats = cfg.njoints, 1
        self.precision = precision
        self.device_index = device
        self.max_batch = max_batch
        self._sched_key = None
        self._loaded = set()
        self._n_params = 0
        self._source = _clone_of          # keeps the weight owner alive as long as any lane exists
        if _clone_of is not None:
            h = C.c_void_p()
            self.lib.check(self.lib.cdll.dsg_clone(_clone_of.handle, max_batch, C.byref(h)))
            self.handle = h
            self._n_params = _clone_of._n_params
            return
        c = L.dsg_config()
        c.variant, c.njoints, c.n_poses, c.n_seed = cfg.variant, cfg.njoints, cfg.n_poses, cfg.n_seed
        c.latent_dim, c.audio_src_dim, c.audio_dim = cfg.latent_dim, cfg.audio_src_dim, cfg.audio_dim
        c.style_dim_in, c.window, c.num_layers = cfg.style_dim_in, cfg.window, cfg.num_layers
        c.num_heads, c.ff_size, c.local_heads = cfg.num_heads, cfg.ff_size, cfg.local_heads
        c.pe_max_len, c.train_steps, c.max_batch = cfg.pe_max_len, 1000, max_batch
        c.precision = {"fp32": L.PREC_FP32, "bf16": L.PREC_BF16, "bf16w2": L.PREC_BF16W2}[precision]
        c.device, c.steps_per_graph = device, steps_per_graph
        c.latency_mode = {"auto": 0, "off": 1, "on": 2}[latency_mode]
        h = C.c_void_p()
        self.lib.check(self.lib.cdll.dsg_create(C.byref(c), C.byref(h)))
        self.handle = h

    def clone(self, max_batch: int | None = None) -> "DSGDenoiser":
        """A further sampling lane over the SAME device weights (dsg_clone): own stream / HSA queue, state, conditioning,
        schedule.  One lane per concurrently sampled clip; `DSGDiffusion.p_sample_loop_multi` advances lanes together."""
        return DSGDenoiser(self.cfg, self.precision, max_batch or self.max_batch, self.device_index, library=self.lib,
                           _clone_of=self._source or self)

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
        """One placeholder tensor on the model's device, sized like the checkpoint's parameter count: the callers only ask
        `next(model.parameters()).device` and `sum(p.numel() for p in model.parameters())` (the weights themselves live
        repacked inside the library)."""
        import torch
        dev = torch.device(f"cuda:{self.device_index}") if torch.cuda.is_available() else torch.device("cpu")
        yield torch.empty(1, device=dev).expand(self._n_params) if self._n_params else torch.empty(0, device=dev)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Feeds every tensor to dsg_load_tensor under its checkpoint key, then repacks (dsg_finalize_weights).
        Unexpected keys raise like `load_model_wo_clip` asserts (model_util.py:11); `clip_model.*` keys are ignored
        the way the reference tolerates them as missing (model_util.py:12)."""
        self._n_params = 0
        for name, t in state_dict.items():
            if name.startswith("clip_model."):
                continue
            b = L.Buf(t)
            if not (name.endswith(".pe") or name.endswith("inv_freq")):      # buffers are not parameters
                self._n_params += int(np.prod(b.obj.shape))
            shape = tuple(int(s) for s in b.obj.shape)
            arr = (C.c_int64 * len(shape))(*shape)
            self.lib.check(self.lib.cdll.dsg_load_tensor(self.handle, name.encode(), b.p, arr, len(shape), 0))
            self._loaded.add(name)
        self.lib.check(self.lib.cdll.dsg_finalize_weights(self.handle))
        return [], []

    def set_schedule(self, diffusion):
        betas = np.ascontiguousarray(diffusion.betas, dtype=np.float64)
        tmap = np.ascontiguousarray(diffusion.timestep_map, dtype=np.int64)
        key = (betas.tobytes(), tmap.tobytes())          # by content: object ids are recycled
        if self._sched_key == key:
            return
        self.lib.check(self.lib.cdll.dsg_set_schedule(self.handle, betas.ctypes.data, tmap.ctypes.data, len(betas)))
        self._sched_key = key

    def set_cond(self, y: dict, batch: int, uncond: bool = False, cfg_scale=None):
        """Per-window conditioning (the `y` dict of model_kwargs).  `cfg_scale` [batch]: classifier-free guidance fused
        into the path (dsg_set_window_cond_cfg; needs max_batch >= 2 * batch)."""
        style, seed, audio = L.Buf(y["style"]), L.Buf(y.get("seed")), L.Buf(y["audio"])
        mask = y["mask_local"]         # KeyError when absent, like the reference's y['mask_local'] (mdm.py:214); None = `mask=None`
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
        if cfg_scale is not None:
            if uncond:
                raise ValueError("guidance and y['uncond'] exclude each other")
            sc = L.Buf(cfg_scale)
            if int(np.prod(sc.obj.shape)) != batch:
                raise ValueError(f"y['scale'] must have {batch} entries")
            self.lib.check(self.lib.cdll.dsg_set_window_cond_cfg(self.handle, style.p, seed.p, audio.p, mbuf.p, mb, batch,
                                                                 sc.p, stream))
            return
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

    def forward(self, x, timesteps, y=None, uncond_info=False, *, cfg_scale=None):
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
        self.set_cond(y, B, uncond=uncond_info, cfg_scale=cfg_scale)
        out, optr = self._alloc_out(xb.obj.shape, use_torch)
        self.lib.check(self.lib.cdll.dsg_forward(self.handle, xb.p, tb.p, optr, B,
                                                 L.current_stream_ptr() if use_torch else None))
        return out

    __call__ = forward

    def sync(self):
        self.lib.check(self.lib.cdll.dsg_sync(self.handle))

    def last_sample_path(self) -> str:
        """How the step loop of the last sampling call was submitted: "hip" launches, hand-written "aql" packets or "graph"
        replays."""
        p = C.c_int()
        self.lib.check(self.lib.cdll.dsg_last_sample_path(self.handle, C.byref(p)))
        return {0: "hip", 1: "aql", 2: "graph"}[p.value]

    # ---- kernel sets (include/dsg.h DSG_KSET_*) -------------------------------------------------------------------
    def set_kernel_set(self, name: str):
        """Which hand-written kernels a denoising step of this lane is made of: "auto" (by batch), "latency", "tile", "block".
        Sticky; `clone()`s made afterwards inherit it.  Sets differ in the last bits (bf16): a lane reproduces another run bit
        for bit only under the same set."""
        self.lib.check(self.lib.cdll.dsg_set_kernel_set(self.handle, L.KERNEL_SETS[name]))
        return self

    def kernel_set(self) -> str:
        """The set in force for this lane ("auto" unless `set_kernel_set` / DSG_KSET chose one)."""
        p = C.c_int()
        self.lib.check(self.lib.cdll.dsg_get_kernel_set(self.handle, C.byref(p)))
        return L.KERNEL_SET_NAMES[p.value]

    def recommend_kernel_set(self, batch: int, lanes: int = 1) -> str:
        """The set measured fastest for `lanes` lanes of `batch` clips advanced together (dsg_recommend_kernel_set)."""
        p = C.c_int()
        self.lib.check(self.lib.cdll.dsg_recommend_kernel_set(self.handle, int(batch), int(lanes), C.byref(p)))
        return L.KERNEL_SET_NAMES[p.value]

    def last_kernel_set(self) -> str:
        """The set the last forward / sampling call of this lane ran."""
        p = C.c_int()
        self.lib.check(self.lib.cdll.dsg_last_kernel_set(self.handle, C.byref(p)))
        return L.KERNEL_SET_NAMES[p.value]

    def last_sample_fence_free(self) -> bool:
        """True when the AQL packets of the last sampling call's loop carried no acquire / release fences (loop-written buffers
        in uncached memory: the default for max_batch <= 16; DSG_UC=0 turns it off)."""
        p = C.c_int()
        self.lib.check(self.lib.cdll.dsg_last_sample_fence_free(self.handle, C.byref(p)))
        return bool(p.value)

    def last_sample_ms(self):
        ms, n = C.c_float(), C.c_int()
        self.lib.check(self.lib.cdll.dsg_last_sample_ms(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class ClassifierFreeSampleModel:
    """Classifier-free guidance wrapper, sampling only (main/model/cfg_sampler.py:8-31): `out_uncond + y['scale'] * (out -
    out_uncond)`, the unconditional evaluation with `y['uncond'] = True` (which zeroes the style embedding -- and, for
    DiffuseStyleGesture, the seed-pose embedding input -- mdm.py:156-164, :180).  The reference class asserts `cond_mode in
    ['text', 'action']` (inherited from MDM) and therefore cannot wrap the gesture models at all; this one can.

    Wrapping a `DSGDenoiser` keeps everything inside the library: the conditional rows and their unconditional twins run as
    ONE batch of 2B rows and the pose-head epilogue combines them (dsg_set_window_cond_cfg) -- in `forward` and, through
    `DSGDiffusion.p_sample_loop` / `ddim_sample_loop`, in the fused step loop with the framework's Philox noise.  The
    wrapped denoiser needs `max_batch >= 2 * batch`; with less, `forward` falls back to two library calls."""

    def __init__(self, model):
        self.model = model
        self.cfg, self.njoints, self.nfeats = model.cfg, model.njoints, model.nfeats

    def parameters(self):
        return self.model.parameters()

    def forward(self, x, timesteps, y=None):
        if y is None or "scale" not in y:
            raise KeyError("scale")
        scale = y["scale"]
        B = int(x.shape[0])
        if isinstance(self.model, DSGDenoiser) and self.model.max_batch >= 2 * B:
            return self.model.forward(x, timesteps, {k: v for k, v in y.items() if k != "scale"}, cfg_scale=scale)
        y_uncond = dict(y)
        y_uncond["uncond"] = True
        out = self.model(x, timesteps, y)
        out_uncond = self.model(x, timesteps, y_uncond)
        scale = scale.view(-1, 1, 1, 1) if L.is_torch(scale) else np.asarray(scale, np.float32).reshape(-1, 1, 1, 1)
        return out_uncond + scale * (out - out_uncond)

    __call__ = forward
