"""MI355X parity tests (-m gpu), round 2: the full config[1] workload end to end (.bvh parity, N1), config[2] as a combination
(DDIM-50 at batch 16), the DSG+ callers against the reference's own `inference()` (G11), TWH chains, the submission path
assertions, sampling lanes ("one clip per stream"), fused guidance, clip_denoised, `mask=None`, both command lines.
Tolerances: rel-L2 on normalised poses as in test_gpu_parity.py; BVH channel error in degrees / cm stated per test."""
import os

import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

TOL_FWD = {"fp32": 2e-5, "bf16": 1.2e-2}     # bf16: <= 2x the 4.5e-3 .. 6.7e-3 measured on MI355X
TOL_CHAIN = {"fp32": 1e-4, "bf16": 2e-2}     # bf16: <= 2x the 9.3e-3 measured after 1000 steps


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from diffusestylegesture_amd import lib as L
    return L.default_library()


def _model(cfg, prec, max_batch=1, wseed=20240, spg=0, latency_mode="auto"):
    from diffusestylegesture_amd.model import DSGDenoiser
    m = DSGDenoiser(cfg, precision=prec, max_batch=max_batch, device=0, steps_per_graph=spg, latency_mode=latency_mode)
    m.load_state_dict(synth_state_dict(cfg, wseed))
    return m


# ---------------------------------------------------------------------------------------------------------------------
# N1: "outputs match the reference .bvh frames within a stated L2 tolerance on identical seeds" -- the whole config[1]
# workload (4 windows x 1000 steps) through de-normalisation and the BVH writer, against the reference's inference() +
# pose2bvh driven with the same Philox noise (G12 poses, G13 .bvh channels).
#   stated tolerance, fp32 kernels: rel-L2 <= 1e-5 on de-normalised poses; BVH rotations max <= 2e-3 deg, root position <= 1e-3 cm
#       (measured on MI355X: 2.0e-7; 1.1e-4 deg; 2.7e-5 cm)
#   stated tolerance, bf16 kernels: rel-L2 <= 2e-2 on NORMALISED poses (the per-window bound; windows are chained);
#       BVH rotation channels max <= 1.5 deg, median <= 0.03 deg; root position <= 0.7 cm   (~2x the measured values)
#       (measured: 9.3e-3; max 0.72 / p99 0.26 / median 0.011 deg; 0.32 cm)
#   stated tolerance, bf16w2 kernels (round 5; hi + lo bf16 weights and GEMM operands): rel-L2 <= 1.3e-3 (normalised), rotation max <= 0.09 deg,
#       median <= 1.5e-3 deg, root position <= 0.025 cm   (2x the measured 6.4e-4; 0.045 / 7.3e-4 deg; 0.012 cm)
# (the measured values are printed by the test and recorded in DESIGN.md s2 / profiles/)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16w2"])
def test_full_clip_1000_steps_bvh_parity(gpu, golden_dir, tmp_path, prec):
    import torch
    from diffusestylegesture_amd import bvh
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import denormalise, generate_clip
    g12, g13 = _g(golden_dir, "g12_clip1000_zeggs.npz"), _g(golden_dir, "g13_bvh1000_zeggs.npz")
    ms = _g(golden_dir, "zeggs_mean_std.npz")
    cfg = C.ZEGGS
    m = _model(cfg, prec, wseed=int(g12["wseed"]))
    d = create_gaussian_diffusion()
    feats = [torch.from_numpy(synth_window_inputs(cfg, 1, window=w)["audio"]).cuda() for w in range(4)]
    poses = generate_clip(m, d, feats, [1, 0, 0, 0, 0, 0], seed=int(g12["noise_seed"]), smoothing=True, skip_timesteps=0)[0]
    assert poses.shape == (312, 1141) and np.isfinite(poses).all()
    assert d.last_sample_path() == "aql"
    ref_den = g12["poses_denorm"].astype(np.float64)
    std = np.clip(ms["std"], 0.01, None)
    ref_norm = (ref_den - ms["mean"]) / std
    e_norm, e_den = rel_l2(poses, ref_norm), rel_l2(denormalise(poses, ms["mean"], ms["std"]), ref_den)
    p = str(tmp_path / f"clip_{prec}.bvh")
    bvh.pose2bvh(poses, p, 312, True, mean=ms["mean"], std=ms["std"])
    vals = np.array([[float(v) for v in r.split()] for r in open(p).read().split("MOTION\n")[1].strip().split("\n")[2:]])
    ref = g13["motion_smooth"].astype(np.float64)
    assert vals.shape == ref.shape == (936, 228)
    dpos = np.abs(vals[:, :3] - ref[:, :3])
    drot = np.abs(vals[:, 3:] - ref[:, 3:])
    drot = np.minimum(drot, np.abs(drot - 360.0))
    stats = dict(rel_l2_norm=e_norm, rel_l2_denorm=e_den, rot_max=drot.max(), rot_p99=np.percentile(drot, 99), rot_median=np.median(drot),
                 pos_max_cm=dpos.max(), pos_median_cm=np.median(dpos))
    print(f"N1 {prec}: " + " ".join(f"{k}={v:.3e}" for k, v in stats.items()))
    if prec == "fp32":
        assert e_den < 1e-5 and drot.max() < 2e-3 and dpos.max() < 1e-3, stats
    elif prec == "bf16w2":       # round 5: weights + the step's own GEMM operands as hi + lo bf16 (measured: 6.4e-4; max 0.045 / median 7.3e-4 deg; 0.012 cm)
        assert e_norm < 1.3e-3 and drot.max() < 0.09 and np.median(drot) < 1.5e-3 and dpos.max() < 0.025, stats
    else:
        assert e_norm < 2e-2 and drot.max() < 1.5 and np.median(drot) < 0.03 and dpos.max() < 0.7, stats


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ddim50_batch16_distinct_rows_vs_oracle(gpu, prec):
    """config[2] as a combination: DDIM-50 at batch 16, every row its own conditioning and its own Philox noise, vs the
    oracle run row by row (rows are independent)."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, B = C.ZEGGS, 16
    m = _model(cfg, prec, max_batch=B)
    y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    d50 = create_gaussian_diffusion("ddim50").manual_seed(77, 5)
    s = np.asarray(d50.ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, eta=0.0))
    assert np.isfinite(s).all()
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    od = OracleDiffusion(timestep_respacing="ddim50")
    rows = [0, 7, 15] if prec == "bf16" else [0, 5, 10, 15]
    for b in rows:
        yb = {k: (v[b:b + 1] if k != "mask_local" else v) for k, v in y.items()}
        nf = lambda k, b=b: philox.normal_bj1t(shape, 77, k, 5)[b:b + 1]
        r = sampler.ddim_sample_loop(od, ref, (1,) + shape[1:], nf, {"y": yb}, eta=0.0)
        assert rel_l2(s[b], r[0]) < TOL_CHAIN[prec], (b, rel_l2(s[b], r[0]))
    assert not np.array_equal(s[0], s[1])


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_twh_chain_and_clip_vs_oracle(gpu, prec):
    """config[4] dims (TWH, latent 512: the un-fused kernel set): 12-step DDPM chain and a 3-window clip vs the oracle"""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip_dsgplus
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.TWH
    sd = synth_state_dict(cfg, 20240)
    m = _model(cfg, prec)
    ref, od = MDMOracle(sd, cfg), OracleDiffusion()
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=2, seed_pose_scale=0.1)
    d = create_gaussian_diffusion().manual_seed(13, 2)
    s = d.p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=988)
    r = sampler.p_sample_loop(od, ref, shape, sampler.philox_noise_fn(shape, 13, 2), {"y": y}, skip_timesteps=988)
    assert rel_l2(s, r) < TOL_CHAIN[prec]
    feats = [synth_window_inputs(cfg, 1, window=w)["audio"] for w in range(3)]
    seed0 = synth_window_inputs(cfg, 1, window=0, seed_pose_scale=0.1)["seed"]
    style = [0.0] * cfg.style_dim_in
    style[3] = 1.0
    got = generate_clip_dsgplus(m, d, feats, style, seed0, 290, seed=5, skip_timesteps=997)

    def sample_window(c, yy):
        nf = lambda k: philox.normal_bj1t(shape, 5, c * 4 + k, 0)
        return sampler.p_sample_loop(od, ref, shape, nf, {"y": yy}, skip_timesteps=997)
    want = sampler.dsgplus_clip(sample_window, cfg, feats, style, seed0, 290)
    assert got.shape == (1, 290, cfg.njoints // 3)
    assert rel_l2(got[0], want) < TOL_CHAIN[prec]


@pytest.mark.parametrize("name,cfg", [("DiffuseStyleGesture+", C.BEAT), ("DiffuseStyleGesture++", C.BEATPP), ("DiffuseStyleGesture", C.BEAT3)])
def test_dsgplus_callers_vs_reference_inference(gpu, golden_dir, name, cfg):
    """the HIP path under the DSG+ clip driver vs the reference's own inference() for its three model names (G11)"""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip_dsgplus
    from diffusestylegesture_amd.sample_plus import seed_features, window_features
    from tests.conftest import ROOT
    g = _g(golden_dir, "g11_clip_dsgplus.npz")
    ms = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "beat_twh_mean_std.npz"))
    mean, std = ms["BEAT_mean"], ms["BEAT_std"]
    real_n = int(g["real_n_frames"])
    ta = np.concatenate([synth_window_inputs(C.BEAT, 1, window=w)["audio"][0] for w in range(3)])[:real_n]
    wins, _ = window_features(ta, 0, cfg.stride)
    seed0 = seed_features(g["seed_raw"], mean, std)
    for prec in ("fp32", "bf16"):
        m = _model(cfg, prec, wseed=int(g["wseed"]))
        d = create_gaussian_diffusion()
        seq = generate_clip_dsgplus(m, d, [w[None] for w in wins], [1.0, 0.0], seed0, real_n, seed=int(g["noise_seed"]),
                                    skip_timesteps=int(g["skip_timesteps"]), seed_last=seed0 if cfg.variant == 5 else None)[0]
        out = np.multiply(seq, std) + mean
        e = rel_l2(out, g[name])
        print(f"G11 {name} {prec}: {e:.3e}")
        assert e < (2e-5 if prec == "fp32" else 1.2e-2), (name, prec, e)


def test_attention3_beat_dims_vs_reference(gpu, golden_dir):
    g = _g(golden_dir, "g14_forward_attn3_beat.npz")
    for cfg in (C.BEAT3, C.TINY3B):
        B, _, rs, ts = (int(v) for v in g[cfg.name + "_meta"])
        y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
        x = np.random.RandomState(rs).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        for prec in ("fp32", "bf16"):
            m = _model(cfg, prec, max_batch=B, wseed=int(g["wseed"]))
            assert rel_l2(m(x, np.array([ts] * B), y), g[cfg.name + "_out"]) < TOL_FWD[prec]


def test_submission_path_is_what_was_asked_for(gpu, monkeypatch):
    """dsg_last_sample_path: the AQL / HIP-launch / hipGraph comparisons elsewhere are only meaningful if each mode ran"""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    cfg = C.ZEGGS
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=2)
    outs = {}
    for env, spg, want in (("1", 0, "aql"), ("0", 0, "hip"), ("1", 10, "graph")):
        monkeypatch.setenv("DSG_AQL", env)
        m = _model(cfg, "bf16", spg=spg)
        d = create_gaussian_diffusion().manual_seed(5, 1)
        outs[want] = np.asarray(d.p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=950)).copy()
        assert m.last_sample_path() == want == d.last_sample_path()
    assert np.array_equal(outs["aql"], outs["hip"]) and np.array_equal(outs["aql"], outs["graph"])


def test_graph_replay_follows_skip_and_schedule_changes(gpu):
    """steps_per_graph > 0: a captured graph bakes the step-table length in; a later call with MORE steps, or another
    schedule, must not replay it (round-1 advisor finding)."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    cfg = C.ZEGGS
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=1)
    mg, me = _model(cfg, "bf16", spg=8), _model(cfg, "bf16", spg=-1)
    for resp, skips in (("", (980, 940, 990)), ("ddim50", (30, 0, 40))):
        for skip in skips:
            res = []
            for m in (mg, me):
                d = create_gaussian_diffusion(resp).manual_seed(8, 0)
                fn = d.ddim_sample_loop if resp else d.p_sample_loop
                res.append(np.asarray(fn(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=skip)).copy())
            assert np.array_equal(res[0], res[1]), (resp, skip)
    assert mg.last_sample_path() == "graph" and me.last_sample_path() in ("aql", "hip")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_lanes_one_clip_per_stream(gpu, prec):
    """dsg_clone / dsg_sample_multi on the hardware: 4 lanes over one weight copy, own HSA queues, interleaved step loops --
    bit-identical to sampling each clip alone, and the AQL path must be the one that ran"""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip, generate_clips_streams
    cfg = C.ZEGGS
    m = _model(cfg, prec)
    lanes = [m] + [m.clone() for _ in range(3)]
    d = create_gaussian_diffusion()
    feats = [[torch.from_numpy(synth_window_inputs(cfg, 1, window=w, clip0=c)["audio"]).cuda() for w in range(2)] for c in range(4)]
    got = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=31, skip_timesteps=940, stream_ids=[0, 1, 2, 3])
    assert all(ln.last_sample_path() == "aql" for ln in lanes)
    for c in range(4):
        want = generate_clip(m, d, feats[c], [1, 0, 0, 0, 0, 0], seed=31, skip_timesteps=940, stream_id=c)
        assert np.array_equal(got[c], want[0]), c
    assert not np.array_equal(got[0], got[1])


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_guidance_fused_vs_oracle(gpu, golden_dir, prec):
    """classifier-free guidance inside the step loop (2B rows, combined in the pose-head epilogue, Philox noise) vs the
    oracle with two evaluations per step, scale in {0.5, 2.5}; forward vs the composition of the reference goldens"""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import ClassifierFreeSampleModel
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    gt = _g(golden_dir, "gt_tiny_zeggs.npz")
    m = _model(C.TINY, prec, max_batch=4, wseed=int(gt["wseed"]))
    y = synth_window_inputs(C.TINY, 2, window=2, seed_pose_scale=0.3)
    x = np.random.RandomState(99).randn(2, C.TINY.njoints, 1, C.TINY.n_poses).astype(np.float32)
    sc = np.array([2.5, 0.5], np.float32)
    want = gt["fwd_uncond"] + sc.reshape(-1, 1, 1, 1) * (gt["fwd_allones"] - gt["fwd_uncond"])
    w = ClassifierFreeSampleModel(m)
    assert rel_l2(w(x, np.array([998, 17]), dict(y, scale=sc)), want) < 3 * TOL_FWD[prec]
    cfg = C.ZEGGS
    mz = _model(cfg, prec, max_batch=2)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    yz = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.3)
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    for scale in (0.5, 2.5):
        yy = dict(yz, scale=np.array([scale], np.float32))
        d = create_gaussian_diffusion().manual_seed(3, 9)
        s = d.p_sample_loop(ClassifierFreeSampleModel(mz), shape, clip_denoised=False, model_kwargs={"y": yy}, skip_timesteps=990)
        r = sampler.p_sample_loop(OracleDiffusion(), sampler.CFGModel(ref), shape, sampler.philox_noise_fn(shape, 3, 9), {"y": yy},
                                  skip_timesteps=990)
        assert rel_l2(s, r) < 2 * TOL_CHAIN[prec], (scale, rel_l2(s, r))


def test_clip_denoised_mask_none_and_generic_loop_noise(gpu):
    """clip_denoised=True in the fused epilogue; y['mask_local'] = None; the generic loop (any callable) draws the SAME Philox
    stream as the fused loop, so a wrapped model reproduces the fused DDPM result"""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.TINY
    m = _model(cfg, "fp32", max_batch=2)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    y = synth_window_inputs(cfg, 2, window=0, seed_pose_scale=0.4)
    shape = (2, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion()
    s = d.manual_seed(5, 1).p_sample_loop(m, shape, model_kwargs={"y": y}, skip_timesteps=990)
    r = sampler.p_sample_loop(OracleDiffusion(), ref, shape, sampler.philox_noise_fn(shape, 5, 1), {"y": y}, skip_timesteps=990,
                              clip_denoised=True)
    assert rel_l2(s, r) < 1e-4
    x = np.random.RandomState(1).randn(*shape).astype(np.float32)
    yn = dict(y, mask_local=None)
    assert rel_l2(m(x, np.array([500, 20]), yn), ref(x, [500, 20], yn)) < 2e-5
    yt = {k: torch.from_numpy(v).cuda() for k, v in y.items()}

    class Wrapped:       # not a DSGDenoiser -> generic loop
        def __call__(self, xx, tt, y=None):
            return m(xx, tt, y)

        def parameters(self):
            return m.parameters()
    fused = d.manual_seed(7, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=985)
    gen = d.manual_seed(7, 2).p_sample_loop(Wrapped(), shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=985)
    assert rel_l2(gen.cpu().numpy(), fused.cpu().numpy()) < 1e-5
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(v.shape)) for k, v in synth_state_dict(cfg, 20240).items()
                                                        if not (k.endswith(".pe") or k.endswith("inv_freq")))


def test_command_lines_end_to_end(gpu, tmp_path):
    """f3: checkpoint file -> `sample.main([...])` -> .bvh, equal to the API path; `sample_plus.main` likewise (poses .npy)"""
    import torch
    from diffusestylegesture_amd import bvh, sample, sample_plus
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from tests.conftest import ROOT
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    ck = str(tmp_path / "model000450000.pt")
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)
    feats = np.stack([synth_window_inputs(cfg, 1, window=w)["audio"][0] for w in range(2)])
    fn = str(tmp_path / "015_Happy_4_x_1_0.npy")
    np.save(fn, feats)
    out = sample.main(["--model_path", ck, "--features_npy", fn, "--save_dir", str(tmp_path / "out"), "--gpu", "0",
                       "--timestep_respacing", "ddim50"])
    assert out.endswith("015_Happy_4_x_1_0.bvh") and os.path.getsize(out) > 500000
    m = _model(cfg, "bf16")
    d = create_gaussian_diffusion("ddim50")
    poses = sample.generate_clip(m, d, [torch.from_numpy(f[None]).cuda() for f in feats], sample.style2onehot["Happy"], seed=123456)[0]
    assert np.array_equal(poses, np.load(out.replace(".bvh", "_poses.npy")))
    ms = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "zeggs_mean_std.npz"))
    ref = str(tmp_path / "ref.bvh")
    bvh.pose2bvh(sample.denormalise(poses, ms["mean"], ms["std"]), ref, poses.shape[0], True)
    assert open(ref).read() == open(out).read()
    # DSG+ / DSG++ / attention3 command line at BEAT dims, 2 windows, 3 steps each
    msb = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "beat_twh_mean_std.npz"))
    rs = np.random.RandomState(3)
    seed_raw = msb["BEAT_mean"] + msb["BEAT_std"] * 0.5 * rs.randn(C.BEAT.n_seed + 2, msb["BEAT_mean"].shape[-1])
    np.save(str(tmp_path / "seed.npy"), seed_raw)
    ta = np.concatenate([synth_window_inputs(C.BEAT, 1, window=w)["audio"][0] for w in range(2)])[:200]
    np.save(str(tmp_path / "ta.npy"), ta)
    for name, c in (("DiffuseStyleGesture+", C.BEAT), ("DiffuseStyleGesture++", C.BEATPP), ("DiffuseStyleGesture", C.BEAT3)):
        ckb = str(tmp_path / f"{c.name}.pt")
        torch.save({k: torch.from_numpy(v) for k, v in synth_state_dict(c, 20240).items()}, ckb)
        argv = ["--model_path", ckb, "--features_npy", str(tmp_path / "ta.npy"), "--seed_npy", str(tmp_path / "seed.npy"), "--name", name,
                "--save_dir", str(tmp_path / ("o" + c.name)), "--skip_timesteps", "997", "--dataset", "BEAT"]
        if c.variant == 5:
            argv += ["--seed_last_npy", str(tmp_path / "seed.npy")]
        res = np.load(sample_plus.main(argv))
        assert res.shape == (200, c.njoints // 3) and np.isfinite(res).all()


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3])
def test_fence_free_loop_equals_fenced_loop(monkeypatch, batch):
    """Default submission: loop-written buffers in uncached memory, AQL packets without acquire / release (dsg_hip.cpp uc_mode).
    Same kernels, same arithmetic as DSG_UC=0 (cached buffers, agent-scope fences): the samples must be bit-identical -- a
    stale line in any cache would show."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import DSGDenoiser
    cfg = C.ZEGGS
    shape = (batch, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion()
    ms = {}
    for uc in ("0", "1"):
        monkeypatch.setenv("DSG_UC", uc)
        ms[uc] = DSGDenoiser(cfg, precision="bf16", max_batch=batch)
        ms[uc].load_state_dict(synth_state_dict(cfg, 1))
    for w in range(2):
        y = {"y": synth_window_inputs(cfg, batch, window=w, seed_pose_scale=0.2)}
        outs = {uc: np.asarray(d.manual_seed(21, w).p_sample_loop(ms[uc], shape, clip_denoised=False, model_kwargs=y, skip_timesteps=600))
                for uc in ("0", "1")}
        assert ms["1"].last_sample_path() == "aql" and ms["0"].last_sample_path() == "aql"
        assert ms["1"].last_sample_fence_free() and not ms["0"].last_sample_fence_free()
        assert np.array_equal(outs["0"], outs["1"])
