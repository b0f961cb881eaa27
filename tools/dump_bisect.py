#!/usr/bin/env python3
"""Dump internal buffers after one 1-layer forward (bf16) -- run under the emulator (DSG_LIB=...emu.so) and on the GPU, then diff."""
import ctypes as C, os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diffusestylegesture_amd import config as CF, lib as L
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
name, lm, outp = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = dataclasses.replace(CF.CONFIGS[name], num_layers=1)
lib = L.DSGLibrary(os.environ.get("DSG_LIB")) if os.environ.get("DSG_LIB") else L.default_library()
m = DSGDenoiser(cfg, precision="bf16", max_batch=1, device=0, latency_mode=lm, library=lib)
m.load_state_dict(synth_state_dict(cfg, 20240))
y = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.1)
x = np.random.RandomState(5).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
o = m(x, np.array([500]), y)
fn = lib.cdll.dsg_debug_read
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]
res = {"out": o}
bf = {"X0a", "q", "k", "vt", "attn", "hidden", "xsA"}
for nm in ["xsA", "partial", "X0", "X0a", "q", "k", "vt", "attn", "pre1", "X1", "hidden", "pre2", "fwd_out"]:
    buf = np.zeros(64 << 20, np.uint8); n = C.c_longlong()
    rc = fn(m.handle, nm.encode(), buf.ctypes.data, buf.nbytes, C.byref(n))
    if rc: print(nm, "rc", rc); continue
    raw = buf[: n.value]
    if nm in bf:
        a = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    else:
        a = raw.view(np.float32)
    res[nm] = a.copy()
    print(f"{nm:8s} n={a.size:8d} finite={bool(np.isfinite(a).all())} nan={int(np.isnan(a).sum())} absmean={float(np.abs(np.nan_to_num(a)).mean()):.5f}")
np.savez_compressed(outp, **res)
