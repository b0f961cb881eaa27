"""ctypes binding of libdsg_hip.so (C ABI: include/dsg.h).

The product path has exactly one backend: the HIP library built in-tree by `make` / `__graft_entry__.build()`.
If it is missing or fails to load, importing a model raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libdsg_hip.so")

E_INVALID, E_RUNTIME, E_UNEXPECTED_KEY, E_MISSING_KEY, E_NOT_IMPLEMENTED, E_STATE = -1, -2, -3, -4, -5, -6
PREC_FP32, PREC_BF16, PREC_BF16W2 = 0, 1, 2
MODE_DDPM, MODE_DDIM = 0, 1
KERNEL_SETS = {"auto": 0, "latency": 1, "tile": 2, "block": 3, "stream": 4, "rows": 5}          # DSG_KSET_* of include/dsg.h
KERNEL_SET_NAMES = {v: k for k, v in KERNEL_SETS.items()}


class dsg_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "variant", "njoints", "n_poses", "n_seed", "latent_dim", "audio_src_dim", "audio_dim", "style_dim_in",
        "window", "num_layers", "num_heads", "ff_size", "local_heads", "pe_max_len", "train_steps", "max_batch",
        "precision", "device", "steps_per_graph", "latency_mode")] + [("reserved", C.c_int32 * 4)]


class dsg_sample_args(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("skip_timesteps", C.c_int32), ("eta", C.c_float), ("const_noise", C.c_int32),
        ("init_noise", C.c_void_p), ("step_noise", C.c_void_p), ("init_image", C.c_void_p),
        ("seed", C.c_uint64), ("stream_id", C.c_uint64), ("draw_base", C.c_uint32), ("n_dump", C.c_int32),
        ("dump_steps", C.c_void_p), ("dump_out", C.c_void_p), ("clip_denoised", C.c_int32), ("first_step", C.c_int32), ("max_steps", C.c_int32), ("reserved", C.c_int32 * 1)]


# every symbol include/dsg.h declares: name -> (restype, argtypes)
_P, _I, _I64 = C.c_void_p, C.c_int, C.c_int64
SYMBOLS = {
    "dsg_version": (_I, []),
    "dsg_last_error": (C.c_char_p, []),
    "dsg_create": (_I, [C.POINTER(dsg_config), C.POINTER(_P)]),
    "dsg_clone": (_I, [_P, _I, C.POINTER(_P)]),
    "dsg_destroy": (_I, [_P]),
    "dsg_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _I]),
    "dsg_finalize_weights": (_I, [_P]),
    "dsg_set_schedule": (_I, [_P, _P, _P, _I]),
    "dsg_schedule_tables": (_I, [_P, _I, _P]),
    "dsg_set_seed_last": (_I, [_P, _P, _I, _P]),
    "dsg_set_window_cond": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dsg_set_window_cond_cfg": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "dsg_forward": (_I, [_P, _P, _P, _P, _I, _P]),
    "dsg_sample": (_I, [_P, C.POINTER(dsg_sample_args), _P, _I, _P]),
    "dsg_sample_multi": (_I, [C.POINTER(_P), _I, C.POINTER(dsg_sample_args), C.POINTER(_P), _I, _P]),
    "dsg_set_kernel_set": (_I, [_P, _I]),
    "dsg_get_kernel_set": (_I, [_P, C.POINTER(_I)]),
    "dsg_recommend_kernel_set": (_I, [_P, _I, _I, C.POINTER(_I)]),
    "dsg_last_kernel_set": (_I, [_P, C.POINTER(_I)]),
    "dsg_sync": (_I, [_P]),
    "dsg_last_sample_ms": (_I, [_P, C.POINTER(C.c_float), C.POINTER(_I)]),
    "dsg_last_sample_path": (_I, [_P, C.POINTER(_I)]),
    "dsg_last_sample_fence_free": (_I, [_P, C.POINTER(_I)]),
    "dsg_trim": (_I, [_I, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dsg_noise": (_I, [_P, _I, _I, _I, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "dsg_pose2bvh": (_I, [_P, _I, _I, _P, _P, _I, C.c_char_p]),
    "dsg_pose2bvh_channels": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "dsg_pose2bvh_batch": (_I, [_P, _I, _I, _I, _P, _P, _I, C.POINTER(C.c_char_p)]),
    "dsg_q_sample": (_I, [_P, _P, _P, _P, _P, _I, _I64, _P]),
    "dsg_predict_xstart_from_eps": (_I, [_P, _P, _P, _P, _P, _I, _I64, _P]),
    "dsg_posterior_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I64, _P]),
    "dsg_ddim_step": (_I, [_P, _P, _P, _P, _P, _I, _I64, _P]),
}


class DSGError(RuntimeError):
    pass


class DSGLibrary:
    def __init__(self, path: str | None = None):
        path = path or os.environ.get("DSG_LIB", DEFAULT_LIB)
        if not os.path.exists(path):
            raise DSGError(
                f"{path} not found: build the HIP library first (`make` or `python -c 'import __graft_entry__ as g; "
                f"g.build()'`).  There is no CPU fallback for the sampling path.")
        self.path = path
        # torch (when installed) must own the HIP runtime of the process: the library exchanges hipStream_t handles and
        # device pointers with it, and a second libamdhip64 initialised first leaves the later one without devices
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.cdll = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self.cdll, name)          # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        if self.cdll.dsg_version() < 330:
            raise DSGError("libdsg_hip.so is older than this package")

    def check(self, rc: int):
        if rc == 0:
            return
        msg = (self.cdll.dsg_last_error() or b"").decode(errors="replace")
        if rc in (E_INVALID, E_UNEXPECTED_KEY, E_MISSING_KEY):
            raise ValueError(msg)
        if rc == E_NOT_IMPLEMENTED:
            raise NotImplementedError(msg)
        raise DSGError(f"[{rc}] {msg}")


def trim(device: int = -1, library: "DSGLibrary | None" = None):
    """dsg_trim: hand the uncached arenas that hold no live block back to HIP; returns (bytes released, bytes still held)."""
    lib = library or default_library()
    rel, held = C.c_longlong(0), C.c_longlong(0)
    lib.check(lib.cdll.dsg_trim(device, C.byref(rel), C.byref(held)))
    return int(rel.value), int(held.value)


_default = None


def default_library() -> DSGLibrary:
    global _default
    if _default is None:
        _default = DSGLibrary()
    return _default


# ---- tensor plumbing (torch is only a carrier for device memory; numpy works too) -------------------------------
def is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class Buf:
    """Keeps a contiguous fp32 (or given dtype) view alive and exposes its address."""

    def __init__(self, x, dtype="float32"):
        if x is None:
            self.obj, self.ptr = None, None
            return
        if is_torch(x):
            import torch
            td = {"float32": torch.float32, "int64": torch.int64, "uint8": torch.uint8}[dtype]
            if dtype == "uint8" and x.dtype == torch.bool:
                x = x.to(torch.uint8)
            self.obj = x.detach().to(td).contiguous()
            self.ptr = self.obj.data_ptr()
        else:
            a = np.asarray(x)
            if dtype == "uint8" and a.dtype == np.bool_:
                a = a.astype(np.uint8)
            self.obj = np.ascontiguousarray(a, dtype=dtype)
            self.ptr = self.obj.ctypes.data

    @property
    def p(self):
        return C.c_void_p(self.ptr) if self.ptr is not None else None


def current_stream_ptr(device_index=None):
    """hipStream_t of torch's current stream (0 / None when torch has no GPU)."""
    try:
        import torch
        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)
    except Exception:
        pass
    return None
