"""Host-side pieces of the two command lines (no GPU): the reference's flags parse, audio / feature windowing and the
DSG+ seed-gesture features follow `sample.py` (main/mydiffusion_zeggs/sample.py:214-249, BEAT-TWH-main/.../sample.py:52-73,
:112-129)."""
import numpy as np

from diffusestylegesture_amd import sample, sample_plus


def test_zeggs_flags_and_audio_windows():
    a = sample.build_parser().parse_args(["--config", "c.yml", "--gpu", "0", "--model_path", "m.pt", "--audiowavlm_path",
                                          "015_Happy_4_x_1_0.wav", "--max_len", "320"])
    assert a.max_len == 320 and a.audiowavlm_path.endswith(".wav")
    wav = np.arange(16000 * 9, dtype=np.float32)
    wins, n = sample.window_audio(wav, 0)
    assert n == 160 and len(wins) == 2 and wins[0].shape == (88 * 800,)
    assert not wins[0][: 8 * 800].any() and np.array_equal(wins[0][8 * 800:], wav[: 80 * 800])      # zero left context
    assert np.array_equal(wins[1][: 8 * 800], wav[72 * 800: 80 * 800])                               # previous chunk's tail
    assert sample.style2onehot["Happy"] == [1, 0, 0, 0, 0, 0]


def test_dsgplus_flags_windows_and_seed_features():
    a = sample_plus.build_parser().parse_args(["--dataset", "TWH", "--tst_prefix", "x", "--skip_timesteps", "3",
                                               "--features_npy", "f.npy", "--seed_npy", "s.npy"])
    assert a.dataset == "TWH" and a.skip_timesteps == 3
    ta = np.random.RandomState(0).randn(250, 7).astype(np.float32)
    w, n = sample_plus.window_features(ta, 0, 120)
    assert n == 250 and w.shape == (3, 120, 7) and np.array_equal(w.reshape(-1, 7)[:250], ta) and not w.reshape(-1, 7)[250:].any()
    w1, n1 = sample_plus.window_features(ta, 100, 120)
    assert n1 == 100 and w1.shape == (1, 120, 7)
    g = np.random.RandomState(1).randn(8, 5)
    mean, std = np.full(5, 0.5), np.full(5, 2.0)
    f = sample_plus.seed_features(g, mean, std)
    gn = (g - mean) / std
    assert f.shape == (1, 15, 1, 6)
    assert np.allclose(f[0, :5, 0, :].T, gn[2:], atol=1e-6)
    assert np.allclose(f[0, 5:10, 0, :].T, (gn[1:] - gn[:-1])[1:], atol=1e-6)
    assert np.allclose(f[0, 10:, 0, :].T, gn[2:] - 2 * gn[1:-1] + gn[:-2], atol=1e-6)


def test_bench_cpu_baseline_leg_runs_on_cpu():
    """bench.py's cpu_baseline leg (the numpy oracle timed on the host cores) must work without a GPU and report what it used."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
        out = mod.cpu_baseline(20)
    finally:
        sys.argv = argv
    assert out["kind"].startswith("port") and out["unit"] == "frames/s" and out["value"] > 0 and out["cores"] >= 1
    assert "20 DDPM steps" in out["sample"]
