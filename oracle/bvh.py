"""TEST INFRASTRUCTURE (oracle): pose vector (ZEGGS, 1141-d) -> BVH, the step right after the sampling path.
The product writer is C++ (diffusestylegesture_amd/csrc/dsg_bvh.cpp behind dsg_pose2bvh); this numpy restatement is only the
checker it is compared with (tests/test_bvh.py) and is itself pinned to the reference writer's own text (G7).

Own numpy restatement of the reference's post-processing chain
  * `pose2bvh`                          main/process/process_zeggs_bvh.py:219-275 (slices, Savitzky-Golay 15/2, x3 repeat)
  * `xform_orthogonalize_from_xy`       ubisoft-laforge-ZeroEGGS-main/ZEGGS/anim/txform.py:23-34 (fp32, eps 1e-10)
  * `quat.from_xform/mul/mul_vec/to_euler`   .../anim/quat.py:166-206, :17-40, :111-120
  * `write_bvh`                         .../ZEGGS/utils_zeggs.py:47-87 (root composed into joint 0 AFTER the repeat)
  * `bvh.save` / `save_joint`           .../anim/bvh.py:137-234 (text layout, `%f`, DFS order, End Sites)
"""
from __future__ import annotations

import numpy as np

PARENTS = np.array([-1, 0, 1, 2, 3, 4, 5, 6, 7, 4, 9, 10, 11, 12, 13, 14, 15, 12, 17, 18, 19, 12, 21, 22, 23, 12, 25, 26,
                    27, 12, 29, 30, 31, 12, 11, 4, 35, 36, 37, 38, 39, 40, 41, 38, 43, 44, 45, 38, 47, 48, 49, 38, 51,
                    52, 53, 38, 55, 56, 57, 38, 37, 0, 61, 62, 63, 64, 63, 62, 0, 68, 69, 70, 71, 70, 69], dtype=np.int32)

_FINGERS = ["Thumb", "Index", "Middle", "Ring", "Pinky"]


def _bone_names():
    names = ["Hips", "Spine", "Spine1", "Spine2", "Spine3", "Neck", "Neck1", "Head", "HeadEnd"]
    for side in ("Right", "Left"):
        names += [side + "Shoulder", side + "Arm", side + "ForeArm", side + "Hand"]
        for f in _FINGERS:
            names += [f"{side}Hand{f}{i}" for i in range(1, 5)]
        names += [side + "ForeArmEnd", side + "ArmEnd"]
    for side in ("Right", "Left"):
        names += [side + "UpLeg", side + "Leg", side + "Foot", side + "ToeBase", side + "ToeBaseEnd", side + "LegEnd",
                  side + "UpLegEnd"]
    return names


BONE_NAMES = _bone_names()
NJOINTS = 75
assert len(BONE_NAMES) == NJOINTS == len(PARENTS)


# ---- quaternion helpers (w, x, y, z) -------------------------------------------------------------------------------
def _cross(a, b):
    o = np.empty(np.broadcast(a, b).shape)
    o[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    o[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    o[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return o


def quat_mul(x, y):
    x0, x1, x2, x3 = (x[..., i:i + 1] for i in range(4))
    y0, y1, y2, y3 = (y[..., i:i + 1] for i in range(4))
    return np.concatenate([y0 * x0 - y1 * x1 - y2 * x2 - y3 * x3, y0 * x1 + y1 * x0 - y2 * x3 + y3 * x2,
                           y0 * x2 + y1 * x3 + y2 * x0 - y3 * x1, y0 * x3 - y1 * x2 + y2 * x1 + y3 * x0], axis=-1)


def quat_mul_vec(q, v):
    t = 2.0 * _cross(q[..., 1:], v)
    return v + q[..., 0][..., None] * t + _cross(q[..., 1:], t)


def quat_to_euler_zyx(x):
    x0, x1, x2, x3 = (x[..., i:i + 1] for i in range(4))
    return np.concatenate([np.arctan2(2.0 * (x0 * x3 + x1 * x2), 1.0 - 2.0 * (x2 * x2 + x3 * x3)),
                           np.arcsin(np.clip(2.0 * (x0 * x2 - x3 * x1), -1.0, 1.0)),
                           np.arctan2(2.0 * (x0 * x1 + x2 * x3), 1.0 - 2.0 * (x1 * x1 + x2 * x2))], axis=-1)


def quat_from_xform(ts, eps=1e-10):
    """Rotation matrices [..., 3, 3] -> quaternions, branch on the largest diagonal term (quat.py:166-206)."""
    m = lambda i, j: ts[..., i, j]
    t = m(0, 0) + m(1, 1) + m(2, 2)
    out = np.empty(ts.shape[:-2] + (4,), dtype=ts.dtype)
    s = 0.5 / np.sqrt(np.maximum(t + 1, eps))
    c = t > 0
    q_t = np.stack([0.25 / s, s * (m(2, 1) - m(1, 2)), s * (m(0, 2) - m(2, 0)), s * (m(1, 0) - m(0, 1))], -1)
    c0 = (m(0, 0) > m(1, 1)) & (m(0, 0) > m(2, 2))
    s0 = 2.0 * np.sqrt(np.maximum(1.0 + m(0, 0) - m(1, 1) - m(2, 2), eps))
    q0 = np.stack([(m(2, 1) - m(1, 2)) / s0, s0 * 0.25, (m(0, 1) + m(1, 0)) / s0, (m(0, 2) + m(2, 0)) / s0], -1)
    c1 = (~c0) & (m(1, 1) > m(2, 2))
    s1 = 2.0 * np.sqrt(np.maximum(1.0 + m(1, 1) - m(0, 0) - m(2, 2), eps))
    q1 = np.stack([(m(0, 2) - m(2, 0)) / s1, (m(0, 1) + m(1, 0)) / s1, s1 * 0.25, (m(1, 2) + m(2, 1)) / s1], -1)
    s2 = 2.0 * np.sqrt(np.maximum(1.0 + m(2, 2) - m(0, 0) - m(1, 1), eps))
    q2 = np.stack([(m(1, 0) - m(0, 1)) / s2, (m(0, 2) + m(2, 0)) / s2, (m(1, 2) + m(2, 1)) / s2, s2 * 0.25], -1)
    out[...] = q2
    out = np.where((c1 & ~c)[..., None], q1, out)
    out = np.where((c0 & ~c)[..., None], q0, out)
    out = np.where(c[..., None], q_t, out)
    return out.astype(ts.dtype)


def orthogonalize_from_xy(xy, eps=1e-10):
    """[..., 2, 3] (x axis, y hint) -> rotation matrix whose COLUMNS are the normalised x, y = z*x, z = x*y axes; fp32."""
    xy = np.asarray(xy, dtype=np.float32)
    xa = xy[..., 0, :]
    za = np.cross(xa, xy[..., 1, :]).astype(np.float32)
    ya = np.cross(za, xa).astype(np.float32)
    n = lambda v: (v / (np.sqrt(np.sum(v * v, -1, dtype=np.float32))[..., None] + np.float32(eps))).astype(np.float32)
    rows = np.stack([n(xa), n(ya), n(za)], axis=-2)
    return np.swapaxes(rows, -1, -2)


# ---- pose vector -> channels -----------------------------------------------------------------------------------------
def pose_to_channels(poses, length, smoothing=False):
    """poses [length, 1141] de-normalised -> (offsets [75,3], positions [3*length,75,3], euler_deg [3*length,75,3])."""
    from scipy.signal import savgol_filter
    poses = np.asarray(poses, dtype=np.float64)
    out = savgol_filter(poses, 15, 2, axis=0) if smoothing else poses
    nj = NJOINTS
    root_pos, root_rot = out[:, 0:3], out[:, 3:7]
    lpos = out[:, 13: 13 + nj * 3].reshape(length, nj, 3)
    ltxy = out[:, 13 + nj * 3: 13 + nj * 9].reshape(length, nj, 2, 3)
    lrot = quat_from_xform(orthogonalize_from_xy(ltxy))
    rep = lambda a: np.repeat(a, 3, axis=0)                    # 20 fps -> 60 fps
    root_pos, root_rot, lpos, lrot = rep(root_pos), rep(root_rot), rep(lpos).copy(), rep(lrot).astype(np.float64)
    lpos[:, 0] = quat_mul_vec(root_rot, lpos[:, 0]) + root_pos
    lrot[:, 0] = quat_mul(root_rot, lrot[:, 0])
    return lpos[0].copy(), lpos, np.degrees(quat_to_euler_zyx(lrot))


def _hierarchy(offsets):
    lines, jseq = [], [0]
    t = ""
    lines.append("HIERARCHY")
    lines.append("ROOT %s" % BONE_NAMES[0])
    lines.append("{")
    t = "\t"
    lines.append("%sOFFSET %f %f %f" % ((t,) + tuple(offsets[0])))
    lines.append("%sCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation " % t)

    def joint(i, t):
        jseq.append(i)
        lines.append("%sJOINT %s" % (t, BONE_NAMES[i]))
        lines.append("%s{" % t)
        t2 = t + "\t"
        lines.append("%sOFFSET %f %f %f" % ((t2,) + tuple(offsets[i])))
        lines.append("%sCHANNELS 3 Zrotation Yrotation Xrotation" % t2)
        kids = [j for j in range(NJOINTS) if PARENTS[j] == i]
        for j in kids:
            joint(j, t2)
        if not kids:
            lines.append("%sEnd Site" % t2)
            lines.append("%s{" % t2)
            lines.append("%s\tOFFSET %f %f %f" % (t2, 0.0, 0.0, 0.0))
            lines.append("%s}" % t2)
        lines.append("%s}" % t)

    for i in range(NJOINTS):
        if PARENTS[i] == 0:
            joint(i, t)
    lines.append("}")
    return lines, jseq


def pose2bvh(poses, outpath, length, smoothing=False):
    """Same call as the reference's `pose2bvh(poses, outpath, length, smoothing)`; writes the .bvh text file."""
    offsets, positions, rots = pose_to_channels(poses, length, smoothing)
    lines, jseq = _hierarchy(offsets)
    n = rots.shape[0]
    lines.append("MOTION")
    lines.append("Frames: %i" % n)
    lines.append("Frame Time: %f" % (1 / 60))
    with open(outpath, "w") as f:
        f.write("\n".join(lines) + "\n")
        for i in range(n):
            parts = []
            for j in jseq:
                if j == 0:
                    parts.append("%f %f %f %f %f %f " % (positions[i, j, 0], positions[i, j, 1], positions[i, j, 2],
                                                         rots[i, j, 0], rots[i, j, 1], rots[i, j, 2]))
                else:
                    parts.append("%f %f %f " % (rots[i, j, 0], rots[i, j, 1], rots[i, j, 2]))
            f.write("".join(parts) + "\n")
    return jseq
