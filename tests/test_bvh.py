"""CPU: the BVH writer against the text the reference's own pose2bvh wrote for the same poses
(tests/golden/g7_bvh_zeggs.npz: hierarchy text + all 936 x 228 motion channel values of the G6 clip; g13: the channels of the
1000-step clip G12).  Product = the C++ writer behind dsg_pose2bvh (csrc/dsg_bvh.cpp); oracle/bvh.py is the numpy checker."""
import os

import numpy as np
import pytest

from diffusestylegesture_amd import bvh
from oracle import bvh as obvh


def _parse(path):
    txt = open(path).read()
    head, motion = txt.split("MOTION\n")
    lines = motion.strip().split("\n")
    vals = np.array([[float(v) for v in r.split()] for r in lines[2:]])
    return head, lines[0], lines[1], vals


def _angle_diff(a, b):
    d = np.abs(a - b)
    return np.minimum(d, np.abs(d - 360.0))          # Euler angles wrap at +-180


@pytest.mark.parametrize("writer", ["cxx", "oracle"])
def test_bvh_matches_reference_writer(golden_dir, tmp_path, hip_lib_path, writer):
    g6 = np.load(os.path.join(golden_dir, "g6_clip_zeggs.npz"))
    g7 = np.load(os.path.join(golden_dir, "g7_bvh_zeggs.npz"))
    poses = g6["poses_denorm"]
    w = bvh.pose2bvh if writer == "cxx" else obvh.pose2bvh
    p = str(tmp_path / "a.bvh")
    w(poses, p, length=312, smoothing=True)
    head, frames, ft, vals = _parse(p)
    assert head == str(g7["header"])                      # hierarchy, joint order, OFFSET lines: identical text
    assert frames == str(g7["frames_line"]) == "Frames: 936" and ft == str(g7["frametime_line"])
    assert vals.shape == (936, 6 + 74 * 3)
    # channel values: degrees / cm printed with 6 decimals; the orthogonalisation runs in fp32 on both sides
    d = _angle_diff(vals, g7["motion_smooth"])
    assert d.max() < 2e-3, d.max()
    assert np.median(d) < 2e-5
    p2 = str(tmp_path / "b.bvh")
    w(poses, p2, length=312, smoothing=False)
    _, _, _, raw = _parse(p2)
    ref = g7["motion_raw_first_last"]
    d = _angle_diff(np.concatenate([raw[:9], raw[-9:]]), ref)
    assert d.max() < 2e-3


def test_cxx_writer_denormalises_batches_and_matches_python_text(golden_dir, tmp_path, hip_lib_path):
    """mean / std path (sample.py:320-326, std clipped at 0.01), float32 input, the batch entry point, the exact `%f`
    formatter against Python's, and the 1000-step clip's channels (G12 -> G13)."""
    ms = np.load(os.path.join(golden_dir, "zeggs_mean_std.npz"))
    g12 = np.load(os.path.join(golden_dir, "g12_clip1000_zeggs.npz"))
    g13 = np.load(os.path.join(golden_dir, "g13_bvh1000_zeggs.npz"))
    den = g12["poses_denorm"].astype(np.float64)
    off, mot = bvh.pose_to_channels(den, 312, smoothing=True)
    d = _angle_diff(mot, g13["motion_smooth"])
    assert d.max() < 2e-3 and np.median(d) < 2e-5
    # normalised float32 poses + mean / std == de-normalised float64 poses
    std = np.clip(ms["std"], 0.01, None)
    norm = ((den - ms["mean"]) / std).astype(np.float32)
    off2, mot2 = bvh.pose_to_channels(norm, 312, smoothing=True, mean=ms["mean"], std=ms["std"])
    assert _angle_diff(mot2, mot).max() < 5e-3            # float32 rounding of the normalised poses
    # text: every value formatted like Python's "%f"
    rs = np.random.RandomState(0)
    clips = np.stack([norm, norm[::-1].copy()])
    paths = [str(tmp_path / f"c{i}.bvh") for i in range(2)]
    bvh.pose2bvh_batch(clips, paths, smoothing=False, mean=ms["mean"], std=ms["std"])
    _, mot_raw = bvh.pose_to_channels(norm, 312, smoothing=False, mean=ms["mean"], std=ms["std"])
    text = open(paths[0]).read().split("MOTION\n")[1].split("\n")[2:]
    for r in rs.randint(0, 936, 40):
        assert text[r] == "".join("%f " % v for v in mot_raw[r])
    one = str(tmp_path / "one.bvh")
    bvh.pose2bvh(norm[::-1].copy(), one, 312, False, mean=ms["mean"], std=ms["std"])
    assert open(one).read() == open(paths[1]).read()
    with pytest.raises(ValueError, match="Savitzky"):
        bvh.pose2bvh(norm[:10], str(tmp_path / "short.bvh"), 10, True)
    with pytest.raises(ValueError):
        bvh.pose2bvh(norm, one, 311, False)


def test_fixed_point_formatter_edge_values(tmp_path, hip_lib_path):
    """the exact `%f` formatter on awkward values: negative zero-rounding, carries, halves, large magnitudes"""
    vals = np.array([0.0, -1e-9, 0.9999995, 0.99999949, 123456.7890125, -0.0000005, 1e-7, 999999.9999996, 2.5e-6, 0.1234565,
                     -179.99999951, 1e8 + 0.1234567], np.float64)
    poses = np.zeros((1, bvh.N_FEATURES), np.float64)
    poses[0, 3] = 1.0                                           # identity root rotation
    poses[0, 13 + 75 * 3: 13 + 75 * 9] = np.tile([1, 0, 0, 0, 1, 0], 75)      # identity joint frames
    for v in vals:
        poses[0, 0] = v                                         # root x position passes straight to the text
        p = str(tmp_path / "v.bvh")
        bvh.pose2bvh(poses, p, 1, False)
        first = open(p).read().split("MOTION\n")[1].split("\n")[2].split()[0]
        assert first == "%f" % v, (v, first)


def test_batch_writer_first_call_from_many_threads(tmp_path, hip_lib_path):
    """the batch writer's worker threads all reach the lazily built skeleton tables at once in a fresh process (a thread-unsafe
    lazy initialisation there crashed `bench.py --clips-per-gpu 16` about every second run)"""
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = (
        "import numpy as np, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from diffusestylegesture_amd import bvh\n"
        "rs = np.random.RandomState(0)\n"
        "p = rs.randn(32, 40, 1141).astype(np.float32)\n"
        "p[..., 3] += 3.0\n"
        f"paths = [{str(tmp_path)!r} + '/c%d.bvh' % i for i in range(32)]\n"
        "bvh.pose2bvh_batch(p, paths, smoothing=True)\n"
        f"bvh.pose2bvh(p[7], {str(tmp_path)!r} + '/one.bvh', 40, True)\n"
        f"assert open(paths[7]).read() == open({str(tmp_path)!r} + '/one.bvh').read()\n"
        "print('ok')\n")
    for _ in range(6):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stderr[-500:])
