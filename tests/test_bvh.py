"""CPU: the BVH writer against the text the reference's own pose2bvh wrote for the same poses
(tests/golden/g7_bvh_zeggs.npz: hierarchy text + all 936 x 228 motion channel values of the G6 clip)."""
import os

import numpy as np

from diffusestylegesture_amd.bvh import pose2bvh


def _parse(path):
    txt = open(path).read()
    head, motion = txt.split("MOTION\n")
    lines = motion.strip().split("\n")
    vals = np.array([[float(v) for v in r.split()] for r in lines[2:]])
    return head, lines[0], lines[1], vals


def test_bvh_matches_reference_writer(golden_dir, tmp_path):
    g6 = np.load(os.path.join(golden_dir, "g6_clip_zeggs.npz"))
    g7 = np.load(os.path.join(golden_dir, "g7_bvh_zeggs.npz"))
    poses = g6["poses_denorm"]
    p = str(tmp_path / "a.bvh")
    pose2bvh(poses, p, length=312, smoothing=True)
    head, frames, ft, vals = _parse(p)
    assert head == str(g7["header"])                      # hierarchy, joint order, OFFSET lines: identical text
    assert frames == str(g7["frames_line"]) == "Frames: 936" and ft == str(g7["frametime_line"])
    assert vals.shape == (936, 6 + 74 * 3)
    # channel values: degrees / cm printed with 6 decimals; the orthogonalisation runs in fp32 on both sides
    d = np.abs(vals - g7["motion_smooth"])
    # Euler angles wrap at +-180: compare modulo 360 on the rotation channels
    d = np.minimum(d, np.abs(d - 360.0))
    assert d.max() < 2e-3, d.max()
    assert np.median(d) < 2e-5
    p2 = str(tmp_path / "b.bvh")
    pose2bvh(poses, p2, length=312, smoothing=False)
    _, _, _, raw = _parse(p2)
    ref = g7["motion_raw_first_last"]
    d = np.abs(np.concatenate([raw[:9], raw[-9:]]) - ref)
    d = np.minimum(d, np.abs(d - 360.0))
    assert d.max() < 2e-3
