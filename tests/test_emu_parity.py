"""CPU: the PRODUCT sources (csrc/dsg_hip.cpp + dsg_kernels.h) compiled for the host under the SIMT emulator
(tests/emu, test infrastructure) and driven through the same ctypes shim, compared with the goldens produced by the
imported reference.  This validates kernel indexing, fragment layouts, host sequencing and the shim before any GPU
minute is spent; the real-hardware parity tests are tests/test_gpu_parity.py (-m gpu)."""
import os

import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.sample import generate_clip, denormalise
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from tests.util import rel_l2

TOL = {"fp32": 1e-5, "bf16": 3e-2}


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope="module")
def tiny(emu_lib, golden_dir):
    gt = _g(golden_dir, "gt_tiny_zeggs.npz")
    sd = synth_state_dict(C.TINY, int(gt["wseed"]))
    models = {}
    for prec in ("fp32", "bf16"):
        m = DSGDenoiser(C.TINY, precision=prec, max_batch=2, library=emu_lib)
        m.load_state_dict(sd)
        models[prec] = m
    y = synth_window_inputs(C.TINY, 2, window=2, seed_pose_scale=0.3)
    x = np.random.RandomState(99).randn(2, C.TINY.njoints, 1, C.TINY.n_poses).astype(np.float32)
    return gt, models, y, x


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_forward_tiny_masks_uncond(tiny, prec):
    gt, models, y, x = tiny
    m = models[prec]
    ts = np.array([998, 17])
    assert rel_l2(m(x, ts, y), gt["fwd_allones"]) < TOL[prec]
    assert rel_l2(m(x, ts, dict(y, mask_local=gt["mask1"])), gt["fwd_mask1"]) < TOL[prec]
    assert rel_l2(m(x, ts, dict(y, mask_local=gt["mask2"])), gt["fwd_mask2"]) < TOL[prec]
    assert rel_l2(m(x, ts, y, uncond_info=True), gt["fwd_uncond"]) < TOL[prec]
    assert rel_l2(m(x, ts, y), gt["fwd_allones"]) < TOL[prec]        # conditioning is re-set per call


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_chains_tiny(tiny, emu_lib, prec):
    gt, models, y, _ = tiny
    m = models[prec]
    shape = (2, C.TINY.njoints, 1, C.TINY.n_poses)
    mk = {"y": y}
    d = create_gaussian_diffusion(library=emu_lib)
    d50 = create_gaussian_diffusion("ddim50", library=emu_lib)
    tol = TOL[prec] * (3 if prec == "fp32" else 1)
    s = d.manual_seed(77, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, skip_timesteps=990)
    assert rel_l2(s, gt["ddpm_skip990"]) < tol
    init = np.random.RandomState(5).randn(*shape).astype(np.float32)
    s = d.manual_seed(77, 4).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, skip_timesteps=992,
                                           init_image=init)
    assert rel_l2(s, gt["ddpm_init_skip992"]) < tol
    s = d.manual_seed(77, 5).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, skip_timesteps=994,
                                           const_noise=True)
    assert rel_l2(s, gt["ddpm_const_noise"]) < tol
    dump = d.manual_seed(77, 6).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, skip_timesteps=994,
                                              dump_steps=[0, 3, 5])
    assert rel_l2(np.stack(dump), gt["ddpm_dump035"]) < tol
    s = d50.manual_seed(77, 8).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk)
    assert rel_l2(s, gt["ddim50_full"]) < tol
    s = d50.manual_seed(77, 9).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, eta=1.0,
                                                skip_timesteps=40)
    assert rel_l2(s, gt["ddim50_eta1_skip40"]) < tol
    # replayed noise (`noise=` and per-step noise pointers) reproduces the Philox-driven run
    from oracle import philox
    n_run = 6
    ext = np.stack([philox.normal_bj1t(shape, 77, 1 + k, 3) for k in range(n_run)])
    x_T = philox.normal_bj1t(shape, 77, 0, 3)
    s1 = d.manual_seed(77, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=mk, skip_timesteps=994)
    s2 = d.p_sample_loop(m, shape, noise=x_T, clip_denoised=False, model_kwargs=mk, skip_timesteps=994, step_noise=ext)
    # (replayed noise runs the pose-space loop, the Philox run the embedded-space loop: same arithmetic in another rounding
    # order -- 1e-6 apart in fp32, two independent bf16 approximations otherwise)
    assert rel_l2(s2, s1) < (1e-5 if prec == "fp32" else 2e-2)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_throughput_kernels_tiny(tiny, emu_lib, prec):
    """The un-fused (batched / throughput) kernel set, which `auto` only selects for batch > 2."""
    gt, _, y, x = tiny
    m = DSGDenoiser(C.TINY, precision=prec, max_batch=2, library=emu_lib, latency_mode="off")
    m.load_state_dict(synth_state_dict(C.TINY, int(gt["wseed"])))
    assert rel_l2(m(x, np.array([998, 17]), dict(y, mask_local=gt["mask2"])), gt["fwd_mask2"]) < TOL[prec]
    d = create_gaussian_diffusion(library=emu_lib)
    shape = (2, C.TINY.njoints, 1, C.TINY.n_poses)
    s = d.manual_seed(77, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990)
    assert rel_l2(s, gt["ddpm_skip990"]) < TOL[prec] * 3
    d50 = create_gaussian_diffusion("ddim50", library=emu_lib)
    s = d50.manual_seed(77, 9).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, eta=1.0,
                                                skip_timesteps=40)
    assert rel_l2(s, gt["ddim50_eta1_skip40"]) < TOL[prec] * 3


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_batch1_attention_fused_into_mid(tiny, emu_lib, prec):
    """Batch 1 takes k_attn_mid (self-attention inside the out_proj/LayerNorm/linear1 kernel).  Same arithmetic and
    rounding points as k_attn + k_mid, so the two must agree bit for bit; and both match the reference goldens."""
    gt, _, y, x = tiny
    y1 = {k: (v[:1] if v.shape[0] == 2 else v) for k, v in y.items()}
    sd = synth_state_dict(C.TINY, int(gt["wseed"]))
    outs = {}
    # fused: batch 1 (k_attn_mid); un-fused: the SAME set at batch 2 (k_attn + k_mid), both rows = the clip (round 6: the DSG_FUSE_ATTN_MID switch is gone)
    for fused, B in (("1", 1), ("0", 2)):
        m = DSGDenoiser(C.TINY, precision=prec, max_batch=B, library=emu_lib, latency_mode="on")
        m.load_state_dict(sd)
        yb = {k: (np.repeat(v[:1], B, 0) if v.shape[0] == 2 else v) for k, v in y.items()}
        outs[fused] = np.asarray(m(np.repeat(x[:1], B, 0), np.array([998] * B), yb))[:1].copy()
        assert rel_l2(outs[fused], gt["fwd_allones"][:1]) < TOL[prec]
        d = create_gaussian_diffusion(library=emu_lib)
        outs["chain" + fused] = np.asarray(d.manual_seed(5, 1).p_sample_loop(
            m, (B, C.TINY.njoints, 1, C.TINY.n_poses), clip_denoised=False, model_kwargs={"y": yb}, skip_timesteps=995))[:1].copy()
    assert np.array_equal(outs["1"], outs["0"])
    assert np.array_equal(outs["chain1"], outs["chain0"])


def test_classifier_free_guidance_wrapper(tiny):
    """cfg_sampler.py:23-31: uncond + scale * (cond - uncond), checked against the reference's two goldens."""
    from diffusestylegesture_amd.model import ClassifierFreeSampleModel
    gt, models, y, x = tiny
    w = ClassifierFreeSampleModel(models["fp32"])
    ts = np.array([998, 17])
    scale = np.array([2.5, 0.5], np.float32)
    want = gt["fwd_uncond"] + scale.reshape(-1, 1, 1, 1) * (gt["fwd_allones"] - gt["fwd_uncond"])
    assert rel_l2(w(x, ts, dict(y, scale=scale)), want) < 2e-5
    with pytest.raises(KeyError):
        w(x, ts, y)


@pytest.mark.parametrize("kset", ["tile", "block", "stream"])
def test_kernel_sets_tiny(tiny, emu_lib, golden_dir, kset):
    """Explicit kernel sets (dsg_set_kernel_set) at the tiny dims against the same goldens as the latency kernels, incl. the
    ragged last row tile (bf16: with k_attn_op; "stream": the weight-stationary GEMMs of dsg_stream.h incl. LayerNorm once per row, the streamed pose embedding and pose head); at batch 8 and batch 23 (529 rows: the 3-waves-per-SIMD LayerNorm GEMMs from
    512 rows) against the oracle; the set that ran is reported (dsg_last_kernel_set) and sticky."""
    from oracle.mdm import MDMOracle
    gt, _, y, x = tiny
    sd = synth_state_dict(C.TINY, int(gt["wseed"]))
    all_precs = ("bf16",) if kset == "stream" else ("fp32", "bf16")       # "stream" (dsg_stream.h, k_ffn) is a bf16 set
    if kset == "stream":
        with pytest.raises(NotImplementedError):
            DSGDenoiser(C.TINY, precision="fp32", max_batch=2, library=emu_lib).set_kernel_set(kset)
    for prec in all_precs:
        m = DSGDenoiser(C.TINY, precision=prec, max_batch=2, library=emu_lib).set_kernel_set(kset)
        m.load_state_dict(sd)
        assert rel_l2(m(x, np.array([998, 17]), dict(y, mask_local=gt["mask2"])), gt["fwd_mask2"]) < TOL[prec]
        assert m.last_kernel_set() == kset
    d = create_gaussian_diffusion(library=emu_lib)
    s = d.manual_seed(77, 3).p_sample_loop(m, (2, C.TINY.njoints, 1, C.TINY.n_poses), clip_denoised=False,
                                           model_kwargs={"y": y}, skip_timesteps=990)
    assert rel_l2(s, gt["ddpm_skip990"]) < TOL["bf16"] and m.last_kernel_set() == kset
    ref = MDMOracle(sd, C.TINY)
    for B, precs in ((8, ("bf16",)), (23, ("fp32", "bf16"))):
        yb = synth_window_inputs(C.TINY, B, window=1, seed_pose_scale=0.3)
        xb = np.random.RandomState(B).randn(B, C.TINY.njoints, 1, C.TINY.n_poses).astype(np.float32)
        ts = np.arange(B) * 40 + 3
        want = ref(xb, list(ts), yb)
        for prec in (p for p in precs if p in all_precs):
            mb = DSGDenoiser(C.TINY, precision=prec, max_batch=B, library=emu_lib).set_kernel_set(kset)
            mb.load_state_dict(sd)
            assert rel_l2(mb(xb, ts, yb), want) < TOL[prec], (B, prec)
    if kset == "tile":
        g2 = _g(golden_dir, "g2_forward_zeggs.npz")
        cfg = C.ZEGGS
        mz = DSGDenoiser(cfg, precision="fp32", max_batch=2, library=emu_lib).set_kernel_set("tile")
        mz.load_state_dict(synth_state_dict(cfg, int(g2["wseed"])))
        yz = synth_window_inputs(cfg, 2, window=1, seed_pose_scale=0.5)
        xz = np.random.RandomState(4244).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        assert rel_l2(mz(xz, np.array([999, 3]), yz), g2["b2_t999_3_out"]) < TOL["fp32"]


def test_kernel_set_is_a_property_of_the_lane_not_of_the_call(tiny, emu_lib):
    """Round-2 advisor finding: the same (handle, batch, seed) must give the same bits whether it is sampled alone or as one
    lane of dsg_sample_multi, also at batch >= 2 per lane -- the kernel set is chosen per handle (explicitly, or by batch),
    never by the number of lanes in the call.  `recommend_kernel_set` is what differs with the lane count."""
    gt, _, _, _ = tiny
    cfg = C.TINY
    m = DSGDenoiser(cfg, precision="bf16", max_batch=2, library=emu_lib)
    m.load_state_dict(synth_state_dict(cfg, int(gt["wseed"])))
    assert m.recommend_kernel_set(1, 1) == "latency" and m.recommend_kernel_set(2, 1) == "latency"
    assert m.recommend_kernel_set(1, 4) == "latency" and m.recommend_kernel_set(2, 4) == "tile"
    assert m.recommend_kernel_set(14, 4) == "rows" and m.recommend_kernel_set(14, 2) == "block" and m.recommend_kernel_set(14, 1) == "tile" and m.recommend_kernel_set(44, 1) == "rows"
    assert m.recommend_kernel_set(130, 1) == "rows" and m.recommend_kernel_set(130, 4) == "stream"      # (round 6: ROWS while one lane's row tiles fit the CUs in one round)
    assert m.recommend_kernel_set(62, 4) == "stream" and m.recommend_kernel_set(62, 1) == "rows"          # ... STREAM once the lanes' row tiles exceed it (tests/test_emu_round6.py has the table)
    shape = (2, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion(library=emu_lib)
    ys = [{"y": synth_window_inputs(cfg, 2, window=w, clip0=2 * w, seed_pose_scale=0.2)} for w in range(2)]
    for kset in ("auto", "tile", "block"):
        lanes = [m.set_kernel_set(kset), m.clone()]
        multi = d.manual_seed(9, 0).p_sample_loop_multi(lanes, shape, ys, seeds=[9, 9], stream_ids=[0, 1], skip_timesteps=994)
        want_set = "latency" if kset == "auto" else kset
        assert all(ln.last_kernel_set() == want_set for ln in lanes)
        for i in range(2):
            alone = d.manual_seed(9, i).p_sample_loop(lanes[i], shape, clip_denoised=False, model_kwargs=ys[i], skip_timesteps=994)
            assert np.array_equal(multi[i], alone), (kset, i)
    with pytest.raises(ValueError):
        emu_lib.check(emu_lib.cdll.dsg_set_kernel_set(m.handle, 9))


def test_error_behaviour(tiny, emu_lib):
    gt, models, y, x = tiny
    m = models["fp32"]
    shape = (2, C.TINY.njoints, 1, C.TINY.n_poses)
    d = create_gaussian_diffusion(library=emu_lib)
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, dump_steps=[1])
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, const_noise=True)
    with pytest.raises(NotImplementedError):
        d.p_sample_loop(m, shape, model_kwargs={"y": y}, cond_fn=lambda *a: None, cond_fn_with_grad=True)       # autograd through the denoiser: not on the path
    with pytest.raises(NotImplementedError):
        d.p_sample_loop(m, shape, model_kwargs={"y": y}, randomize_class=True)          # (denoised_fn / cond_fn run step by step on the device: test_gpu_round5.py)
    with pytest.raises(ValueError):
        d.p_sample_loop(m, (2, 5, 1, 22), clip_denoised=False, model_kwargs={"y": y})
    with pytest.raises(ValueError):
        m(x, np.array([0, 1]), dict(y, audio=y["audio"][:, :5]))
    with pytest.raises(ValueError):
        m(x, np.array([0, 5000]), y)
    m2 = DSGDenoiser(C.TINY, precision="fp32", max_batch=1, library=emu_lib)
    sd = synth_state_dict(C.TINY, 1)
    with pytest.raises(ValueError, match="unexpected key"):
        m2.load_state_dict({"bogus.weight": np.zeros(3, np.float32)})
    sd2 = dict(sd); sd2.pop("input_process2.bias")
    with pytest.raises(ValueError, match="missing key"):
        m2.load_state_dict(sd2)
    with pytest.raises(ValueError, match="size mismatch"):
        m2.load_state_dict({"input_process2.bias": np.zeros(3, np.float32)})
    m2.load_state_dict(dict(sd, **{"clip_model.x": np.zeros(2, np.float32)}))      # tolerated like the reference


def test_forward_tiny4_dsgplus(emu_lib, golden_dir):
    g5 = _g(golden_dir, "g5_forward_dsgplus.npz")
    cfg = C.TINY4
    sd = synth_state_dict(cfg, int(g5["wseed"]))
    y = synth_window_inputs(cfg, 2, window=3, seed_pose_scale=0.1)
    x = np.random.RandomState(33).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    for prec in ("fp32", "bf16"):
        m = DSGDenoiser(cfg, precision=prec, max_batch=2, library=emu_lib)
        m.load_state_dict(sd)
        assert rel_l2(m(x, np.array([500, 500]), y), g5["tiny4_out"]) < TOL[prec]
        assert rel_l2(m(x, np.array([500, 500]), dict(y, uncond=True)), g5["tiny4_uncond"]) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_forward_tiny5_dsgpp(emu_lib, golden_dir, prec):
    """DiffuseStyleGesture++ (variant 5): conditioning rows = [seed embedding | audio | seed_last embedding]."""
    g = _g(golden_dir, "g10_forward_dsgpp.npz")
    cfg, B, ts = C.TINY5, 2, 500
    m = DSGDenoiser(cfg, precision=prec, max_batch=2, library=emu_lib)
    m.load_state_dict(synth_state_dict(cfg, int(g["wseed"])))
    y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
    x = np.random.RandomState(31 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    assert rel_l2(m(x, np.array([ts] * B), y), g["tiny5_out"]) < TOL[prec]
    assert rel_l2(m(x, np.array([ts] * B), dict(y, uncond=True)), g["tiny5_uncond"]) < TOL[prec]
    y_missing = {k: v for k, v in y.items() if k != "seed_last"}
    with pytest.raises(KeyError):
        m(x, np.array([ts] * B), y_missing)


@pytest.mark.parametrize("cfg", [C.TINY4, C.TINY5], ids=["attention4", "attention5"])
def test_dsgplus_clip_tiny_vs_oracle(emu_lib, cfg):
    """DSG+ / DSG++ window loop (ceil windows, seed hand-off, one-frame blend, crop, first third of the features;
    attention5: audio[:-S] per window and the fixed y['seed_last'])."""
    from diffusestylegesture_amd.sample import generate_clip_dsgplus
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    sd = synth_state_dict(cfg, 9)
    m = DSGDenoiser(cfg, precision="fp32", max_batch=1, library=emu_lib)
    m.load_state_dict(sd)
    d = create_gaussian_diffusion(library=emu_lib)
    ref, od = MDMOracle(sd, cfg), OracleDiffusion()
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    feats = [synth_window_inputs(C.TINY4, 1, window=w)["audio"] for w in range(3)]      # stride-long feature windows
    seed0 = synth_window_inputs(cfg, 1, window=0, seed_pose_scale=0.2)["seed"]
    seed_last = synth_window_inputs(cfg, 1).get("seed_last")
    real_n = 61
    got = generate_clip_dsgplus(m, d, feats, [1, 0, 0], seed0, real_n, seed=5, skip_timesteps=996, seed_last=seed_last)
    per = 5

    def sample_window(c, yy):
        nf = lambda k: philox.normal_bj1t(shape, 5, c * per + k, 0)
        return sampler.p_sample_loop(od, ref, shape, nf, {"y": yy}, skip_timesteps=996)
    want = sampler.dsgplus_clip(sample_window, cfg, feats, [1, 0, 0], seed0, real_n, seed_last=seed_last)
    assert got.shape == (1, real_n, cfg.njoints // 3)
    assert rel_l2(got[0], want) < 1e-5


def test_forward_zeggs_full_dims(emu_lib, golden_dir):
    g2 = _g(golden_dir, "g2_forward_zeggs.npz")
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, int(g2["wseed"]))
    m = DSGDenoiser(cfg, precision="fp32", max_batch=2, library=emu_lib)
    m.load_state_dict(sd)
    for name, B, ts, sps in [("b1_t0", 1, [0], 0.0)]:       # (batch 2 at these dims: test_emu_round2 block-GEMM test)
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=sps)
        x = np.random.RandomState(4242 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        assert rel_l2(m(x, np.array(ts), y), g2[name + "_out"]) < TOL["fp32"]
    mb = DSGDenoiser(cfg, precision="bf16", max_batch=1, library=emu_lib)
    mb.load_state_dict(sd)
    y = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.5)
    x = np.random.RandomState(4243).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    assert rel_l2(mb(x, np.array([999]), y), g2["b1_t999_out"]) < TOL["bf16"]


def test_clip_orchestration_matches_reference_inference(emu_lib, golden_dir):
    """generate_clip (window loop, seed hand-off, root shift, one-frame blend, stitching) + de-normalisation vs the
    reference's own inference() output (G6), 4 windows x 3 DDPM steps, one noise stream across windows."""
    g6 = _g(golden_dir, "g6_clip_zeggs.npz")
    ms = _g(golden_dir, "zeggs_mean_std.npz")
    cfg = C.ZEGGS
    m = DSGDenoiser(cfg, precision="fp32", max_batch=1, library=emu_lib)
    m.load_state_dict(synth_state_dict(cfg, int(g6["wseed"])))
    d = create_gaussian_diffusion(library=emu_lib)
    # the first two of the four windows (the emulator is slow at these dims): their 152 emitted frames do not depend on
    # later windows (the 8 overlap frames are cut either by the next window or at the end); all 4 windows run in the GPU suite
    feats = [synth_window_inputs(cfg, 1, window=w)["audio"] for w in range(2)]
    poses = generate_clip(m, d, feats, [1, 0, 0, 0, 0, 0], seed=int(g6["noise_seed"]), smoothing=True,
                          skip_timesteps=int(g6["skip_timesteps"]))
    assert poses.shape == (1, 152, 1141)
    out = denormalise(poses[0], ms["mean"], ms["std"])
    assert rel_l2(out, g6["poses_denorm"][:152]) < 1e-5
