"""Pose vector (ZEGGS, 1141-d) -> BVH: the step right after the sampling path.

`pose2bvh(poses, outpath, length, smoothing)` keeps the reference's call (main/process/process_zeggs_bvh.py:219); the work --
Savitzky-Golay (15, 2), 2-axis orthogonalisation, matrix -> quaternion -> Euler zyx, x3 repeat, the text file -- happens in
the C++ half of libdsg_hip.so (csrc/dsg_bvh.cpp, `dsg_pose2bvh*` of include/dsg.h): ~8 ms per 936-frame file instead of the
1.4 s a Python writer takes, and `pose2bvh_batch` formats many clips on several host threads (the tail of a 128-clip job).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as L

NJOINTS = 75
N_FEATURES = 1141
N_CHANNELS = 6 + (NJOINTS - 1) * 3


def _arr(poses):
    a = np.asarray(poses)
    if a.dtype != np.float32:
        a = a.astype(np.float64, copy=False)
    a = np.ascontiguousarray(a)
    if a.ndim < 2 or a.shape[-1] != N_FEATURES:
        raise ValueError(f"poses must be [..., frames, {N_FEATURES}], got {a.shape}")
    return a, (0 if a.dtype == np.float32 else 1)


def _ms(mean, std):
    if (mean is None) != (std is None):
        raise ValueError("mean and std go together")
    if mean is None:
        return None, None, None, None
    m = np.ascontiguousarray(np.asarray(mean, np.float64).reshape(-1))
    s = np.ascontiguousarray(np.asarray(std, np.float64).reshape(-1))
    if m.shape != (N_FEATURES,) or s.shape != (N_FEATURES,):
        raise ValueError("mean / std must have 1141 entries")
    return m, s, m.ctypes.data, s.ctypes.data


def pose2bvh(poses, outpath, length, smoothing=False, *, mean=None, std=None, library=None):
    """Same call as the reference's `pose2bvh(poses, outpath, length, smoothing)`: de-normalised poses [length, 1141] -> .bvh.
    With `mean` / `std` the poses are the sampler's normalised output and are de-normalised first (sample.py:320-326)."""
    lib = library or L.default_library()
    a, dt = _arr(poses)
    if a.ndim != 2 or a.shape[0] != length:
        raise ValueError(f"poses {a.shape} vs length {length}")      # the reference's reshape([length, ...]) fails likewise
    m, s, mp, sp = _ms(mean, std)
    lib.check(lib.cdll.dsg_pose2bvh(a.ctypes.data, dt, int(length), mp, sp, int(bool(smoothing)), str(outpath).encode()))


def pose_to_channels(poses, length, smoothing=False, *, mean=None, std=None, library=None):
    """(offsets [75, 3], motion [3 * length, 228] in file order) -- the numbers of the file without the text."""
    lib = library or L.default_library()
    a, dt = _arr(poses)
    if a.ndim != 2 or a.shape[0] != length:      # the C++ side reads `length` rows: never let it run past the buffer
        raise ValueError(f"poses {a.shape} vs length {length}")
    m, s, mp, sp = _ms(mean, std)
    off = np.zeros((NJOINTS, 3), np.float64)
    mot = np.zeros((3 * int(length), N_CHANNELS), np.float64)
    lib.check(lib.cdll.dsg_pose2bvh_channels(a.ctypes.data, dt, int(length), mp, sp, int(bool(smoothing)), off.ctypes.data,
                                             mot.ctypes.data))
    return off, mot


def pose2bvh_batch(poses, outpaths, smoothing=False, *, mean=None, std=None, library=None):
    """poses [n_clips, frames, 1141] -> one .bvh per clip, clips spread over host threads."""
    lib = library or L.default_library()
    a, dt = _arr(poses)
    if a.ndim != 3 or a.shape[0] != len(outpaths):
        raise ValueError(f"poses {a.shape} vs {len(outpaths)} paths")
    m, s, mp, sp = _ms(mean, std)
    paths = (C.c_char_p * len(outpaths))(*[str(p).encode() for p in outpaths])
    lib.check(lib.cdll.dsg_pose2bvh_batch(a.ctypes.data, dt, a.shape[0], a.shape[1], mp, sp, int(bool(smoothing)), paths))
