"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the framework's counter-based normal generator.

The reference draws its noise from torch's global generator (`th.randn` at
`main/diffusion/gaussian_diffusion.py:704`, `th.randn_like` at `:542`, `:783`), a stream no HIP
kernel can reproduce.  The framework therefore defines its own stream (Philox4x32-10 + Box-Muller,
implemented in `diffusestylegesture_amd/csrc/dsg_kernels.h: philox_normal4`) and this file restates
it on the CPU so that (a) the oracle sampler and (b) the imported reference (through a patched
`torch.randn/randn_like`, see tests/golden/make_goldens.py) consume the identical noise.

Definition (element e of draw d on stream s, seed k):
    ctr = (e >> 2, d, s & 0xffffffff, s >> 32),  key = (k & 0xffffffff, k >> 32)
    x0..x3 = philox4x32_10(ctr, key)
    u1(x) = ((x >> 8) + 1) * 2^-24 in (0,1],  u2(x) = (x >> 8) * 2^-24 in [0,1)
    z0,z1 = sqrt(-2 ln u1(x0)) * (cos, sin)(2 pi u2(x1));  z2,z3 likewise from (x2, x3);  z = z[e & 3]
Tensor mapping for a [B, J, 1, T] noise tensor: e = (b*T + f) * Jq + j with Jq = 4*ceil(J/4)
(frame-major, so one Philox call feeds four consecutive pose features of one frame).
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  c*: uint32 arrays (broadcastable); k0,k1: uint32 scalars."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape)
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape)
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _box_muller(xa, xb):
    u1 = ((xa >> np.uint32(8)).astype(np.float64) + 1.0) * (2.0 ** -24)
    u2 = (xb >> np.uint32(8)).astype(np.float64) * (2.0 ** -24)
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return r * np.cos(th), r * np.sin(th)


def normal_flat(n_elems: int, seed: int, draw: int, stream: int = 0) -> np.ndarray:
    """z[0:n_elems] (float32) of draw `draw` on `stream` for `seed`."""
    nq = (n_elems + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    c0 = (q & MASK32).astype(np.uint32)
    assert nq < 2 ** 32
    x0, x1, x2, x3 = philox4x32_10(c0, np.uint32(draw & 0xFFFFFFFF), np.uint32(stream & 0xFFFFFFFF),
                                   np.uint32((stream >> 32) & 0xFFFFFFFF),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z0, z1 = _box_muller(x0, x1)
    z2, z3 = _box_muller(x2, x3)
    z = np.stack([z0, z1, z2, z3], axis=1).reshape(-1)[:n_elems]
    return z.astype(np.float32)


def normal_bj1t(shape, seed: int, draw: int, stream: int = 0) -> np.ndarray:
    """Noise tensor of reference shape [B, J, 1, T] under the framework's element mapping."""
    B, J, one, T = shape
    assert one == 1
    Jq = 4 * ((J + 3) // 4)
    z = normal_flat(B * T * Jq, seed, draw, stream).reshape(B, T, Jq)[:, :, :J]
    return np.ascontiguousarray(z.transpose(0, 2, 1))[:, :, None, :]
