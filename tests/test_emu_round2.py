"""CPU (SIMT emulator, test infrastructure): the round-2 additions of the C ABI before any GPU minute is spent --
classifier-free guidance fused into the step (dsg_set_window_cond_cfg), clip_denoised, `mask=None`, sampling lanes over shared
weights (dsg_clone / dsg_sample_multi), the generic loop on the Philox stream (dsg_noise), BEAT-TWH's attention3 model."""
import os

import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import ClassifierFreeSampleModel, DSGDenoiser
from diffusestylegesture_amd.sample import generate_clip, generate_clips_streams
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from oracle import philox, sampler
from oracle.mdm import MDMOracle
from oracle.schedule import OracleDiffusion
from tests.util import rel_l2

TOL = {"fp32": 1e-5, "bf16": 3e-2}


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _model(cfg, prec, lib, max_batch=2, wseed=20240, **kw):
    m = DSGDenoiser(cfg, precision=prec, max_batch=max_batch, library=lib, **kw)
    m.load_state_dict(synth_state_dict(cfg, wseed))
    return m


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_guidance_fused_forward_and_chain(emu_lib, golden_dir, prec):
    """cond + uncond rows as one batch, combined in the pose-head epilogue: forward vs the composition of the two reference
    goldens, a DDPM and a DDIM chain vs the oracle doing two evaluations per step (cfg_sampler.py:8-31)."""
    gt = _g(golden_dir, "gt_tiny_zeggs.npz")
    cfg = C.TINY
    m = _model(cfg, prec, emu_lib, max_batch=4, wseed=int(gt["wseed"]))
    y = synth_window_inputs(cfg, 2, window=2, seed_pose_scale=0.3)
    x = np.random.RandomState(99).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = np.array([998, 17])
    scale = np.array([2.5, 0.5], np.float32)
    want = gt["fwd_uncond"] + scale.reshape(-1, 1, 1, 1) * (gt["fwd_allones"] - gt["fwd_uncond"])
    w = ClassifierFreeSampleModel(m)
    assert rel_l2(w(x, ts, dict(y, scale=scale)), want) < 3 * TOL[prec]
    assert rel_l2(m(x, ts, y), gt["fwd_allones"]) < TOL[prec]            # plain conditioning still works afterwards
    # key masks are shared by the twins
    want_m = None
    ref = MDMOracle(synth_state_dict(cfg, int(gt["wseed"])), cfg)
    y2 = dict(y, mask_local=gt["mask2"])
    want_m = sampler.CFGModel(ref)(x, ts, dict(y2, scale=scale))
    assert rel_l2(w(x, ts, dict(y2, scale=scale)), want_m) < 3 * TOL[prec]
    shape = (2, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion(library=emu_lib)
    s = d.manual_seed(21, 2).p_sample_loop(w, shape, clip_denoised=False, model_kwargs={"y": dict(y, scale=scale)}, skip_timesteps=992)
    r = sampler.p_sample_loop(OracleDiffusion(), sampler.CFGModel(ref), shape, sampler.philox_noise_fn(shape, 21, 2),
                              {"y": dict(y, scale=scale)}, skip_timesteps=992)
    assert rel_l2(s, r) < 3 * TOL[prec]
    d50 = create_gaussian_diffusion("ddim50", library=emu_lib)
    s = d50.manual_seed(22, 1).ddim_sample_loop(w, shape, clip_denoised=False, model_kwargs={"y": dict(y, scale=scale)},
                                                skip_timesteps=44, eta=0.5)
    r = sampler.ddim_sample_loop(OracleDiffusion(timestep_respacing="ddim50"), sampler.CFGModel(ref), shape,
                                 sampler.philox_noise_fn(shape, 22, 1), {"y": dict(y, scale=scale)}, skip_timesteps=44, eta=0.5)
    assert rel_l2(s, r) < 3 * TOL[prec]
    # the fused loop needs room for the twins; a wrapper around a smaller denoiser is handed to the generic loop (two library
    # calls per step, the same Philox stream -- executed on the GPU by test_gpu_round3.py), it does not raise
    small = _model(cfg, prec, emu_lib, max_batch=2, wseed=int(gt["wseed"]))
    assert d._library_model(ClassifierFreeSampleModel(small), 2) == (None, False)
    assert d._library_model(ClassifierFreeSampleModel(small), 1) == (small, True)
    assert d._library_model(small, 2) == (small, False)


def test_guidance_dsgplus_variants(emu_lib):
    """guidance on the DSG+ (attention4) and DSG++ (attention5: the twins share y['seed_last']) models vs the oracle"""
    for cfg in (C.TINY4, C.TINY5):
        m = _model(cfg, "fp32", emu_lib, max_batch=4)
        ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
        y = synth_window_inputs(cfg, 2, window=1, seed_pose_scale=0.2)
        x = np.random.RandomState(3).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        sc = np.array([1.5, 3.0], np.float32)
        got = ClassifierFreeSampleModel(m)(x, np.array([400, 7]), dict(y, scale=sc))
        assert rel_l2(got, sampler.CFGModel(ref)(x, [400, 7], dict(y, scale=sc))) < 3e-5


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_clip_denoised(emu_lib, prec):
    """clip_denoised=True (the reference's default, gaussian_diffusion.py:377-379): x0 clamped before the update"""
    cfg = C.TINY
    m = _model(cfg, prec, emu_lib)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    y = synth_window_inputs(cfg, 2, window=1, seed_pose_scale=0.4)
    shape = (2, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion(library=emu_lib)
    s = d.manual_seed(5, 1).p_sample_loop(m, shape, model_kwargs={"y": y}, skip_timesteps=990)      # clip_denoised defaults to True
    r = sampler.p_sample_loop(OracleDiffusion(), ref, shape, sampler.philox_noise_fn(shape, 5, 1), {"y": y}, skip_timesteps=990,
                              clip_denoised=True)
    r_noclip = sampler.p_sample_loop(OracleDiffusion(), ref, shape, sampler.philox_noise_fn(shape, 5, 1), {"y": y}, skip_timesteps=990)
    assert rel_l2(s, r) < 3 * TOL[prec]
    assert rel_l2(r_noclip, r) > 1e-2, "the case must actually clip"
    d50 = create_gaussian_diffusion("ddim50", library=emu_lib)
    s = d50.manual_seed(6, 1).ddim_sample_loop(m, shape, model_kwargs={"y": y}, skip_timesteps=45)
    r = sampler.ddim_sample_loop(OracleDiffusion(timestep_respacing="ddim50"), ref, shape, sampler.philox_noise_fn(shape, 6, 1),
                                 {"y": y}, skip_timesteps=45, clip_denoised=True)
    assert rel_l2(s, r) < 3 * TOL[prec]


def test_mask_none_and_missing_key(emu_lib):
    """y['mask_local'] = None is LocalAttention's `mask=None`: the look-back pad keys of window 0 attend with value -1
    (local_attention.py:121, :196); a missing key raises KeyError like the reference's dict lookup."""
    cfg = C.TINY
    m = _model(cfg, "fp32", emu_lib)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    y = synth_window_inputs(cfg, 2, window=0, seed_pose_scale=0.3)
    x = np.random.RandomState(1).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = np.array([500, 20])
    yn = dict(y, mask_local=None)
    want = ref(x, ts, yn)
    assert rel_l2(m(x, ts, yn), want) < 1e-5
    assert rel_l2(ref(x, ts, y), want) > 1e-3, "all-ones mask and no mask differ (pads)"
    assert rel_l2(m(x, ts, y), ref(x, ts, y)) < 1e-5
    with pytest.raises(KeyError):
        m(x, ts, {k: v for k, v in y.items() if k != "mask_local"})


def test_lanes_share_weights_and_sample_together(emu_lib):
    """dsg_clone + dsg_sample_multi: lanes over one weight set, advanced together, bit-identical to running each lane alone;
    the clip driver on lanes equals the single-clip driver per Philox stream."""
    cfg = C.TINY
    m = _model(cfg, "fp32", emu_lib, max_batch=1)
    lanes = [m, m.clone(), m.clone()]
    with pytest.raises(RuntimeError):
        lanes[1].load_state_dict(synth_state_dict(cfg, 1))             # clones do not own weights
    d = create_gaussian_diffusion(library=emu_lib)
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    ys = [{"y": synth_window_inputs(cfg, 1, window=w, clip0=w, seed_pose_scale=0.2)} for w in range(3)]
    d.manual_seed(9, 0)
    multi = d.p_sample_loop_multi(list(lanes), shape, ys, seeds=[9, 9, 10], stream_ids=[0, 4, 4], skip_timesteps=993)
    for i, (seed, sid) in enumerate(((9, 0), (9, 4), (10, 4))):
        alone = d.manual_seed(seed, sid).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=ys[i], skip_timesteps=993)
        assert np.array_equal(multi[i], alone), i
    feats = [[synth_window_inputs(cfg, 1, window=w, clip0=c)["audio"] for w in range(3)] for c in range(3)]
    got = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=31, skip_timesteps=996, stream_ids=[0, 1, 2])
    for c in range(3):
        want = generate_clip(m, d, feats[c], [1, 0, 0, 0, 0, 0], seed=31, skip_timesteps=996, stream_id=c)
        assert np.array_equal(got[c], want[0])
    # dropping the owner first is safe: the weights live as long as any lane
    a, b = lanes[1], lanes[2]
    del lanes, m
    out = d.manual_seed(9, 0).p_sample_loop(b, shape, clip_denoised=False, model_kwargs=ys[0], skip_timesteps=993)
    assert np.array_equal(out, multi[0])


@pytest.mark.parametrize("cfgname", ["tiny3b"])
def test_attention3_beat_twh_tree(emu_lib, golden_dir, cfgname):
    """BEAT-TWH-main's cross_local_attention3 model (window 15; name "DiffuseStyleGesture" there) vs the imported reference"""
    g = _g(golden_dir, "g14_forward_attn3_beat.npz")
    cfg = C.CONFIGS[cfgname]
    B, _, rs, ts = (int(v) for v in g[cfgname + "_meta"])
    y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
    x = np.random.RandomState(rs).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    for prec in ("fp32", "bf16"):
        m = _model(cfg, prec, emu_lib, wseed=int(g["wseed"]))
        assert rel_l2(m(x, np.array([ts] * B), y), g[cfgname + "_out"]) < TOL[prec]
        assert rel_l2(m(x, np.array([ts] * B), y, uncond_info=True), g[cfgname + "_uncond"]) < TOL[prec]


def test_block_gemms_of_the_batched_path(emu_lib, golden_dir):
    """Kernel set "block" (dsg_batched.h: 32-row block GEMMs for QKV / linear1 / linear2 / embedding, used from 1000 rows up)
    forced at the small test dims: forward with masks / uncond, DDPM + DDIM chains, guidance, the DSG+ / DSG++ models, and ZEGGS
    dims at batch 2 -- against the same reference goldens as the latency kernels"""
    zeggs_too = True
    gt = _g(golden_dir, "gt_tiny_zeggs.npz")
    cfg = C.TINY
    y = synth_window_inputs(cfg, 2, window=2, seed_pose_scale=0.3)
    x = np.random.RandomState(99).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = np.array([998, 17])
    shape = (2, cfg.njoints, 1, cfg.n_poses)
    for prec in ("fp32", "bf16"):
        m = _model(cfg, prec, emu_lib, max_batch=4, wseed=int(gt["wseed"])).set_kernel_set("block")
        assert rel_l2(m(x, ts, y), gt["fwd_allones"]) < TOL[prec]
        assert rel_l2(m(x, ts, dict(y, mask_local=gt["mask2"])), gt["fwd_mask2"]) < TOL[prec]
        assert rel_l2(m(x, ts, y, uncond_info=True), gt["fwd_uncond"]) < TOL[prec]
        d = create_gaussian_diffusion(library=emu_lib)
        s = d.manual_seed(77, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990)
        assert rel_l2(s, gt["ddpm_skip990"]) < 3 * TOL[prec]
        d50 = create_gaussian_diffusion("ddim50", library=emu_lib)
        s = d50.manual_seed(77, 9).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, eta=1.0, skip_timesteps=40)
        assert rel_l2(s, gt["ddim50_eta1_skip40"]) < 3 * TOL[prec]
        sc = np.array([2.5, 0.5], np.float32)
        want = gt["fwd_uncond"] + sc.reshape(-1, 1, 1, 1) * (gt["fwd_allones"] - gt["fwd_uncond"])
        assert rel_l2(ClassifierFreeSampleModel(m)(x, ts, dict(y, scale=sc)), want) < 3 * TOL[prec]
    g5, g10 = _g(golden_dir, "g5_forward_dsgplus.npz"), _g(golden_dir, "g10_forward_dsgpp.npz")
    for cfg, gold in ((C.TINY4, g5["tiny4_out"]), (C.TINY5, g10["tiny5_out"])):
        yy = synth_window_inputs(cfg, 2, window=3, seed_pose_scale=0.1)
        xx = np.random.RandomState(33).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        m = _model(cfg, "fp32", emu_lib, wseed=int(g5["wseed"])).set_kernel_set("block")
        assert rel_l2(m(xx, np.array([500, 500]), yy), gold) < TOL["fp32"]
    if zeggs_too:
        g2 = _g(golden_dir, "g2_forward_zeggs.npz")
        cfg = C.ZEGGS
        for prec in ("bf16",):           # (fp32 at these dims: test_emu_parity.py::test_kernel_sets_tiny)
            mz = _model(cfg, prec, emu_lib, wseed=int(g2["wseed"])).set_kernel_set("block")
            yz = synth_window_inputs(cfg, 2, window=1, seed_pose_scale=0.5)
            xz = np.random.RandomState(4244).randn(2, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
            assert rel_l2(mz(xz, np.array([999, 3]), yz), g2["b2_t999_3_out"]) < TOL[prec]
