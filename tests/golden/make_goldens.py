#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference (dev container only).

    python tests/golden/make_goldens.py zeggs      # main/ (DiffuseStyleGesture, ZEGGS dims + tiny dims)
    python tests/golden/make_goldens.py dsgplus    # BEAT-TWH-main/ (DiffuseStyleGesture+)
    python tests/golden/make_goldens.py clip       # main/mydiffusion_zeggs/sample.py inference() (G6)
    python tests/golden/make_goldens.py bvh        # main/process/process_zeggs_bvh.py pose2bvh (G7, needs G6)
    python tests/golden/make_goldens.py wavlm      # main/mydiffusion_zeggs/WavLM (G9: small-config feature extractor)
    python tests/golden/make_goldens.py dsgpp      # BEAT-TWH-main/ cross_local_attention5 (DiffuseStyleGesture++, G10)
    python tests/golden/make_goldens.py dsgplus_caller   # BEAT-TWH-main/mydiffusion_beat_twh/sample.py inference() (G11)
    python tests/golden/make_goldens.py clip1000   # G12: inference() at the full 1000 steps per window (config[1] workload)
    python tests/golden/make_goldens.py bvh1000    # G13: the reference .bvh channels of G12
    python tests/golden/make_goldens.py attn3beat  # G14: BEAT-TWH-main cross_local_attention3 (name "DiffuseStyleGesture")
    python tests/golden/make_goldens.py hooks      # G17: p_sample_loop / ddim_sample_loop with denoised_fn and cond_fn (tiny dims)

The two reference trees use the same module names, hence one process per tree.  Nothing from
/root/reference is copied: the script imports it, feeds it seeded synthetic weights / inputs
(diffusestylegesture_amd.synth) and the framework's counter-based noise (oracle.philox, injected by
patching torch.randn / torch.randn_like as seen by the reference sampler), and stores only inputs'
seeds and the reference's outputs.  /root/reference does not exist on the GPU box; tests read the
committed .npz files only.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from oracle import philox

REF = "/root/reference"
WSEED = 20240                 # synthetic-weight seed used by every fixture
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def _to_torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _y_torch(y):
    out = {"style": torch.from_numpy(y["style"]), "seed": torch.from_numpy(y["seed"]),
           "audio": torch.from_numpy(y["audio"]), "mask_local": torch.from_numpy(y["mask_local"])}
    if "seed_last" in y:
        out["seed_last"] = torch.from_numpy(y["seed_last"])
    return out


class NoiseInjector:
    """Replaces torch.randn / torch.randn_like with the framework's Philox stream (draw counter)."""

    def __init__(self, seed, stream=0):
        self.seed, self.stream, self.draw = seed, stream, 0

    def _next(self, shape):
        z = philox.normal_bj1t(tuple(shape), self.seed, self.draw, self.stream)
        self.draw += 1
        return torch.from_numpy(z)

    def __enter__(self):
        self._r, self._rl = torch.randn, torch.randn_like
        torch.randn = lambda *shape, **kw: self._next(shape[0] if isinstance(shape[0], (tuple, list)) else shape)
        torch.randn_like = lambda x, **kw: self._next(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._r, self._rl


def _build_ref_zeggs(cfg):
    from model.mdm import MDM
    m = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode='cross_local_attention3_style1',
            audio_feat='wavlm', arch='trans_enc', latent_dim=cfg.latent_dim, n_seed=cfg.n_seed,
            ff_size=cfg.ff_size, num_layers=cfg.num_layers, num_heads=cfg.num_heads)
    if cfg.pe_max_len != 5000:
        raise SystemExit("reference PositionalEncoding max_len is 5000")
    sd = synth_state_dict(cfg, WSEED)
    missing, unexpected = m.load_state_dict(_to_torch_sd(sd), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval(), sd


def _capture_layers(model, store):
    hooks = []
    for i, layer in enumerate(model.seqTransEncoder.layers):
        hooks.append(layer.register_forward_hook(
            lambda mod, inp, out, i=i: store.__setitem__(f"after_layer{i}", out.permute(1, 0, 2).numpy().copy())))
    return hooks


def gen_zeggs():
    sys.path[:0] = [REF + "/main", REF + "/main/model"]
    np.float = float          # shim: data_loaders/humanml/common/quaternion.py uses np.float
    from utils.model_util import create_gaussian_diffusion
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps

    # ---- G1 schedule tables -------------------------------------------------------------------
    diff = create_gaussian_diffusion()
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
             "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
             "posterior_mean_coef2"]
    g1 = {"full_" + n: np.asarray(getattr(diff, n)) for n in names}
    g1["full_timestep_map"] = np.asarray(diff.timestep_map)

    def spaced_diff(resp):
        betas = gd.get_named_beta_schedule('cosine', 1000, 1.)
        return SpacedDiffusion(use_timesteps=space_timesteps(1000, resp), betas=betas,
                               model_mean_type=gd.ModelMeanType.START_X,
                               model_var_type=gd.ModelVarType.FIXED_SMALL,
                               loss_type=gd.LossType.MSE, rescale_timesteps=False)
    d50 = spaced_diff("ddim50")
    for n in names:
        g1["ddim50_" + n] = np.asarray(getattr(d50, n))
    g1["ddim50_timestep_map"] = np.asarray(d50.timestep_map)
    d3 = spaced_diff("10,15,20")
    g1["sect_timestep_map"] = np.asarray(d3.timestep_map)
    g1["sect_betas"] = np.asarray(d3.betas)
    np.savez_compressed(os.path.join(HERE, "g1_schedule.npz"), **g1)
    print("G1 ok")

    # ---- G2 forward, ZEGGS dims ---------------------------------------------------------------
    cfg = C.ZEGGS
    model, _ = _build_ref_zeggs(cfg)
    g2 = {"wseed": WSEED}
    cases = [("b1_t0", 1, [0], 0.0), ("b1_t999", 1, [999], 0.5), ("b2_t999_3", 2, [999, 3], 0.5)]
    for name, B, ts, sps in cases:
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=sps)
        x = np.random.RandomState(4242 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        store = {}
        hooks = _capture_layers(model, store) if name == "b1_t999" else []
        out = model(torch.from_numpy(x), torch.tensor(ts, dtype=torch.long), y=_y_torch(y)).numpy()
        for h in hooks:
            h.remove()
        g2[name + "_out"] = out.astype(np.float32)
        g2[name + "_meta"] = np.array([B, sps, 4242 + B] + ts, dtype=np.float64)
        if store:
            g2[name + "_after_layer0"] = store["after_layer0"]
            g2[name + f"_after_layer{cfg.num_layers - 1}"] = store[f"after_layer{cfg.num_layers - 1}"]
        print("G2", name, float(np.abs(out).mean()), float(np.abs(out).max()))
    np.savez_compressed(os.path.join(HERE, "g2_forward_zeggs.npz"), **g2)

    # ---- G3 / G4 / G8 sampling chains, ZEGGS dims, B=1 ------------------------------------------
    y = synth_window_inputs(cfg, 1, window=0)
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    g3 = {"wseed": WSEED, "noise_seed": 123456}
    for tag, skip in (("ddpm5", 995), ("ddpm25", 975), ("ddpm1000", 0)):
        with NoiseInjector(123456, stream=0):
            s = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": _y_torch(y)},
                                   skip_timesteps=skip, init_image=None, progress=False,
                                   dump_steps=None, noise=None, const_noise=False)
        g3[tag] = s.numpy().astype(np.float32)
        print("G3", tag, float(np.abs(g3[tag]).mean()))
    for tag, skip, eta in (("ddim50", 0, 0.0), ("ddim5_eta05", 45, 0.5)):
        with NoiseInjector(123456, stream=7):
            s = d50.ddim_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": _y_torch(y)},
                                     skip_timesteps=skip, init_image=None, progress=False, eta=eta)
        g3[tag] = s.numpy().astype(np.float32)
        print("G4", tag, float(np.abs(g3[tag]).mean()))
    np.savez_compressed(os.path.join(HERE, "g3_chains_zeggs.npz"), **g3)

    # ---- tiny dims: masks, batch 2, init_image, const_noise, dump_steps -------------------------
    cfg = C.TINY
    # PositionalEncoding max_len is fixed to 5000 in the reference; the tiny config mirrors that
    model, _ = _build_ref_zeggs(cfg)
    gt = {"wseed": WSEED}
    B = 2
    y = synth_window_inputs(cfg, B, window=2, seed_pose_scale=0.3)
    x = np.random.RandomState(99).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = [998, 17]
    gt["fwd_allones"] = model(torch.from_numpy(x), torch.tensor(ts), y=_y_torch(y)).numpy()
    m1 = np.ones((1, cfg.n_poses), bool); m1[0, [3, 12, 13]] = False
    y1 = dict(y, mask_local=m1)
    gt["mask1"] = m1
    gt["fwd_mask1"] = model(torch.from_numpy(x), torch.tensor(ts), y=_y_torch(y1)).numpy()
    m2 = np.ones((2, cfg.n_poses), bool); m2[0, 0:11] = False; m2[1, [5, 21]] = False
    y2 = dict(y, mask_local=m2)
    gt["mask2"] = m2
    gt["fwd_mask2"] = model(torch.from_numpy(x), torch.tensor(ts), y=_y_torch(y2)).numpy()
    gt["fwd_uncond"] = model(torch.from_numpy(x), torch.tensor(ts), y=_y_torch(y), uncond_info=True).numpy()
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    with NoiseInjector(77, stream=3):
        gt["ddpm_skip990"] = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": _y_torch(y)},
                                                skip_timesteps=990, progress=False).numpy()
    init = np.random.RandomState(5).randn(*shape).astype(np.float32)
    with NoiseInjector(77, stream=4):
        gt["ddpm_init_skip992"] = diff.p_sample_loop(model, shape, clip_denoised=False,
                                                     model_kwargs={"y": _y_torch(y)}, skip_timesteps=992,
                                                     init_image=torch.from_numpy(init), progress=False).numpy()
    with NoiseInjector(77, stream=5):
        gt["ddpm_const_noise"] = diff.p_sample_loop(model, shape, clip_denoised=False,
                                                    model_kwargs={"y": _y_torch(y)}, skip_timesteps=994,
                                                    const_noise=True, progress=False).numpy()
    with NoiseInjector(77, stream=6):
        dump = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": _y_torch(y)},
                                  skip_timesteps=994, dump_steps=[0, 3, 5], progress=False)
    gt["ddpm_dump035"] = np.stack([d.numpy() for d in dump])
    with NoiseInjector(77, stream=8):
        gt["ddim50_full"] = d50.ddim_sample_loop(model, shape, clip_denoised=False,
                                                 model_kwargs={"y": _y_torch(y)}, progress=False, eta=0.0).numpy()
    with NoiseInjector(77, stream=9):
        gt["ddim50_eta1_skip40"] = d50.ddim_sample_loop(model, shape, clip_denoised=False,
                                                        model_kwargs={"y": _y_torch(y)}, progress=False,
                                                        eta=1.0, skip_timesteps=40).numpy()
    with NoiseInjector(77, stream=10):
        gt["ddpm200_tiny"] = diff.p_sample_loop(model, shape, clip_denoised=False,
                                                model_kwargs={"y": _y_torch(y)}, skip_timesteps=800,
                                                progress=False).numpy()
    np.savez_compressed(os.path.join(HERE, "gt_tiny_zeggs.npz"), **gt)
    print("tiny ok", {k: float(np.abs(v).mean()) for k, v in gt.items() if k.startswith(("fwd", "dd"))})


def hook_denoised(x):
    """denoised_fn of G17 (works on torch tensors and numpy arrays): applied to pred_xstart before the clamp (gaussian_diffusion.py:364-370)"""
    return 0.9 * x + 0.01


def hook_cond_scale(t):
    """per-sample factor of G17's cond_fn from the MODEL timestep the wrapped cond_fn receives (respace.py:117-129): t / 1000 + 0.1"""
    return t / 1000.0 + 0.1


def gen_fn_hooks():
    """G17: the sampler hooks of `p_sample` / `ddim_sample` -- denoised_fn (gaussian_diffusion.py:364-370) and cond_fn through condition_mean
    (:428-441, DDPM) / condition_score (:458-480, DDIM) -- at the tiny dims, batch 2.  cond_fn(x, t, y=...) = -5 x (t / 1000 + 0.1)."""
    sys.path[:0] = [REF + "/main", REF + "/main/model"]
    np.float = float
    from utils.model_util import create_gaussian_diffusion
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps
    diff = create_gaussian_diffusion()
    d50 = SpacedDiffusion(use_timesteps=space_timesteps(1000, "ddim50"), betas=gd.get_named_beta_schedule('cosine', 1000, 1.),
                          model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                          loss_type=gd.LossType.MSE, rescale_timesteps=False)
    cfg = C.TINY
    model, _ = _build_ref_zeggs(cfg)
    B = 2
    y = synth_window_inputs(cfg, B, window=2, seed_pose_scale=0.3)
    shape = (B, cfg.njoints, 1, cfg.n_poses)

    def cond_fn(x, t, y=None):
        assert y is not None and "style" in y
        return -5.0 * x * hook_cond_scale(t.float()).view(-1, 1, 1, 1)
    g = {"wseed": WSEED}
    with NoiseInjector(77, stream=11):
        g["ddpm_both_clip_skip800"] = diff.p_sample_loop(model, shape, clip_denoised=True, denoised_fn=hook_denoised, cond_fn=cond_fn,
                                                         model_kwargs={"y": _y_torch(y)}, skip_timesteps=800, progress=False).numpy()
    with NoiseInjector(77, stream=12):
        g["ddpm_denoised_skip992"] = diff.p_sample_loop(model, shape, clip_denoised=False, denoised_fn=hook_denoised,
                                                        model_kwargs={"y": _y_torch(y)}, skip_timesteps=992, progress=False).numpy()
    with NoiseInjector(77, stream=13):
        g["ddpm_cond_skip800"] = diff.p_sample_loop(model, shape, clip_denoised=False, cond_fn=cond_fn,
                                                    model_kwargs={"y": _y_torch(y)}, skip_timesteps=800, progress=False).numpy()    # (200 steps: posterior_variance * grad is ~1e-4 per step up there, nothing at t < 10)
    with NoiseInjector(77, stream=14):
        g["ddim50_both_eta05_skip40"] = d50.ddim_sample_loop(model, shape, clip_denoised=False, denoised_fn=hook_denoised, cond_fn=cond_fn,
                                                             model_kwargs={"y": _y_torch(y)}, progress=False, eta=0.5, skip_timesteps=40).numpy()
    np.savez_compressed(os.path.join(HERE, "g17_sampler_hooks_tiny.npz"), **g)
    print("G17 ok", {k: float(np.abs(v).mean()) for k, v in g.items() if k.startswith("dd")})


def gen_dsgplus():
    sys.path[:0] = [REF + "/BEAT-TWH-main", REF + "/BEAT-TWH-main/model"]
    from model.mdm import MDM
    g5 = {"wseed": WSEED}
    for cfg, ts in ((C.BEAT, 999), (C.TWH, 0), (C.TINY4, 500)):
        m = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode='cross_local_attention4_style1_sample',
                arch='trans_enc', latent_dim=cfg.latent_dim, n_seed=cfg.n_seed, ff_size=cfg.ff_size,
                num_layers=cfg.num_layers, num_heads=cfg.num_heads, style_dim=cfg.style_dim_in,
                source_audio_dim=cfg.audio_src_dim, audio_feat_dim_latent=cfg.audio_dim)
        sd = synth_state_dict(cfg, WSEED)
        if cfg.pe_max_len != 5000:
            raise SystemExit("pe_max_len")
        missing, unexpected = m.load_state_dict(_to_torch_sd(sd), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m.eval()
        B = 1 if cfg.name != "tiny4" else 2
        y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
        x = np.random.RandomState(31 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        out = m(torch.from_numpy(x), torch.tensor([ts] * B), y=_y_torch(y)).numpy()
        g5[cfg.name + "_out"] = out.astype(np.float32)
        g5[cfg.name + "_meta"] = np.array([B, 0.1, 31 + B, ts], dtype=np.float64)
        print("G5", cfg.name, out.shape, float(np.abs(out).mean()))
        if cfg.name == "tiny4":
            yu = _y_torch(y); yu["uncond"] = True
            g5["tiny4_uncond"] = m(torch.from_numpy(x), torch.tensor([ts] * B), y=yu).numpy()
    np.savez_compressed(os.path.join(HERE, "g5_forward_dsgplus.npz"), **g5)


def gen_clip(skip=997, outname="g6_clip_zeggs.npz", f32=False):
    """G6: the reference's own `inference()` (window loop + stitching + de-normalisation) with a fake WavLM
    and 3 DDPM steps per window; pose2bvh is replaced by a capture so no file is written.
    G12 (`clip1000`): the same at the full 1000 steps per window (skip_timesteps=0) -- the config[1] workload."""
    zdir = REF + "/main/mydiffusion_zeggs"
    os.chdir(zdir)
    for name in ("librosa", "omegaconf", "easydict"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["omegaconf"].DictConfig = dict

    class _ED(dict):
        __getattr__ = dict.__getitem__
    sys.modules["easydict"].EasyDict = _ED
    np.float = float
    sys.path[:0] = [zdir]
    import sample as S
    cfg = C.ZEGGS
    S.mydevice = torch.device("cpu")
    S.batch_size = 1
    S.save_dir = "/tmp"
    captured = {}
    S.pose2bvh = lambda poses, path, length, smoothing: captured.__setitem__("poses", np.array(poses))

    n_win = 4
    feats = [synth_window_inputs(cfg, 1, window=w)["audio"] for w in range(n_win)]

    class FakeWavLM:
        def __init__(self):
            self.calls = 0

        def extract_features(self, wav):
            # 219 frames like WavLM on 70400 samples; content chosen so that the reference's
            # align_corners linear interpolation to 88 frames returns feats[w] exactly is not
            # possible -> instead bypass interpolation by patching wav2wavlm below.
            raise RuntimeError
    calls = {"n": 0}

    def fake_wav2wavlm(model, wav, device):
        w = calls["n"]; calls["n"] += 1
        return torch.from_numpy(feats[w])
    S.wav2wavlm = fake_wav2wavlm
    model, _ = _build_ref_zeggs(cfg)
    from utils.model_util import create_gaussian_diffusion
    diff = create_gaussian_diffusion()
    args = _ED(n_poses=88)
    audio = np.zeros(320 * 800, dtype=np.float32)
    style = [1, 0, 0, 0, 0, 0]
    with NoiseInjector(123456, stream=0) as inj:
        # one Philox stream runs through all windows like torch's global generator does (sample.py:212)
        S.inference(args, None, audio, diff.p_sample_loop, model, n_frames=320, smoothing=True,
                    SG_filter=True, minibatch=True, skip_timesteps=skip, style=style, seed=123456)
        draws = inj.draw
    mean = np.load(REF + "/ubisoft-laforge-ZeroEGGS-main/data/processed_v1/processed/mean.npz")["mean"].squeeze()
    std = np.load(REF + "/ubisoft-laforge-ZeroEGGS-main/data/processed_v1/processed/std.npz")["std"].squeeze()
    np.savez_compressed(os.path.join(HERE, outname), poses_denorm=captured["poses"].astype(np.float32 if f32 else np.float64),
                        draws=draws, wseed=WSEED, noise_seed=123456, skip_timesteps=skip)
    np.savez_compressed(os.path.join(HERE, "zeggs_mean_std.npz"), mean=mean, std=std)
    print("G6", captured["poses"].shape, draws)


def gen_bvh(inname="g6_clip_zeggs.npz", outname="g7_bvh_zeggs.npz", full=True):
    """G7: the reference's own pose2bvh (Savitzky-Golay, orthogonalisation, quaternion/Euler, x3 repeat, text writer) on the
    de-normalised poses of G6; stores the hierarchy text and every motion channel value."""
    import tempfile
    sys.modules["omegaconf"] = types.ModuleType("omegaconf")
    sys.modules["omegaconf"].DictConfig = dict
    sys.path[:0] = [REF + "/main/process", REF + "/ubisoft-laforge-ZeroEGGS-main/ZEGGS"]
    os.chdir(REF + "/main/process")
    import process_zeggs_bvh as R
    poses = np.load(os.path.join(HERE, inname))["poses_denorm"].astype(np.float64)
    tmp = tempfile.mkdtemp()
    out = {}
    for sm in (True, False):
        p = os.path.join(tmp, f"ref_{sm}.bvh")
        R.pose2bvh(poses, p, length=312, smoothing=sm)
        head, motion = open(p).read().split("MOTION\n")
        rows = motion.strip().split("\n")
        vals = np.array([[float(v) for v in r.split()] for r in rows[2:]], dtype=np.float64)
        if sm:
            out.update(header=np.array(head), frames_line=np.array(rows[0]), frametime_line=np.array(rows[1]),
                       motion_smooth=vals.astype(np.float32))
        else:
            out["motion_raw_first_last"] = np.concatenate([vals[:9], vals[-9:]]).astype(np.float32)
    if not full:       # G13: only the smoothed motion channels (the header is G7's business)
        out = {"motion_smooth": out["motion_smooth"]}
    np.savez_compressed(os.path.join(HERE, outname), **out)


def gen_attn3_beat():
    """G14: BEAT-TWH-main's `cross_local_attention3_style1_sample` model (name "DiffuseStyleGesture" in that tree:
    BEAT-TWH-main/model/mdm.py:147-185, window 15) at BEAT dims and at tiny dims, conditional + uncond forward."""
    sys.path[:0] = [REF + "/BEAT-TWH-main", REF + "/BEAT-TWH-main/model"]
    from model.mdm import MDM
    g = {"wseed": WSEED}
    for cfg, ts, B in ((C.BEAT3, 640, 1), (C.TINY3B, 500, 2)):
        m = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode='cross_local_attention3_style1_sample',
                arch='trans_enc', latent_dim=cfg.latent_dim, n_seed=cfg.n_seed, ff_size=cfg.ff_size,
                num_layers=cfg.num_layers, num_heads=cfg.num_heads, style_dim=cfg.style_dim_in,
                source_audio_dim=cfg.audio_src_dim, audio_feat_dim_latent=cfg.audio_dim)
        missing, unexpected = m.load_state_dict(_to_torch_sd(synth_state_dict(cfg, WSEED)), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m.eval()
        y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
        x = np.random.RandomState(31 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        out = m(torch.from_numpy(x), torch.tensor([ts] * B), y=_y_torch(y)).numpy()
        g[cfg.name + "_out"] = out.astype(np.float32)
        g[cfg.name + "_meta"] = np.array([B, 0.1, 31 + B, ts], dtype=np.float64)
        if cfg.name == "tiny3b":
            yu = _y_torch(y); yu["uncond"] = True
            g["tiny3b_uncond"] = m(torch.from_numpy(x), torch.tensor([ts] * B), y=yu).numpy()
        print("G14", cfg.name, out.shape, float(np.abs(out).mean()))
    np.savez_compressed(os.path.join(HERE, "g14_forward_attn3_beat.npz"), **g)


def gen_remaining_dims():
    """G15: forward of the imported BEAT-TWH-tree MDM at the remaining name x dataset dims sample.py:297-323 accepts -- TWH under
    "DiffuseStyleGesture" (attention3) and "DiffuseStyleGesture++" (attention5), BEAT "v2" (njoints = 1141) under attention4."""
    sys.path[:0] = [REF + "/BEAT-TWH-main", REF + "/BEAT-TWH-main/model"]
    from model.mdm import MDM
    g = {"wseed": WSEED}
    for cfg, mode, ts in ((C.TWH3, 'cross_local_attention3_style1_sample', 700), (C.TWHPP, 'cross_local_attention5_style1_sample', 12),
                          (C.BEATV2, 'cross_local_attention4_style1_sample', 333)):
        m = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode=mode, arch='trans_enc', latent_dim=cfg.latent_dim, n_seed=cfg.n_seed,
                ff_size=cfg.ff_size, num_layers=cfg.num_layers, num_heads=cfg.num_heads, style_dim=cfg.style_dim_in,
                source_audio_dim=cfg.audio_src_dim, audio_feat_dim_latent=cfg.audio_dim)
        missing, unexpected = m.load_state_dict(_to_torch_sd(synth_state_dict(cfg, WSEED)), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m.eval()
        y = synth_window_inputs(cfg, 1, window=2, seed_pose_scale=0.1)
        x = np.random.RandomState(77).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        out = m(torch.from_numpy(x), torch.tensor([ts]), y=_y_torch(y)).numpy()
        g[cfg.name + "_out"] = out.astype(np.float32)
        g[cfg.name + "_meta"] = np.array([1, 0.1, 77, ts], dtype=np.float64)
        print("G15", cfg.name, out.shape, float(np.abs(out).mean()))
    np.savez_compressed(os.path.join(HERE, "g15_forward_remaining_dims.npz"), **g)


def gen_dsgplus_caller():
    """G11: the reference's own DSG+ caller, `inference()` of BEAT-TWH-main/mydiffusion_beat_twh/sample.py:44-192, driven for
    the three model names it knows (attention3 / 4 / 5) at BEAT dims with 3 DDPM steps per window (skip_timesteps=997) and
    a 300-frame clip (3 windows, zero-padded tail).  Harness-side stand-ins only for what the image lacks and the path does
    not need: `librosa`, `easydict`, the dataset BVH pipelines (`process_BEAT_bvh` / `process_TWH_bvh`: their pose2bvh
    entry points become a capture of `out_poses`), and the ground-truth seed clip (`np.load` of
    ../../BEAT_dataset/...npy returns a seeded synthetic snippet).  Mean / std are the reference's own .npy files."""
    bdir = REF + "/BEAT-TWH-main/mydiffusion_beat_twh"
    os.chdir(bdir)
    captured = {}
    for name in ("librosa", "easydict", "process_BEAT_bvh", "process_TWH_bvh"):
        sys.modules[name] = types.ModuleType(name)

    class _ED(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    sys.modules["easydict"].EasyDict = _ED
    pb, pt = sys.modules["process_BEAT_bvh"], sys.modules["process_TWH_bvh"]
    pb.wav2wavlm = pb.pose2bvh = None
    pb.pose2bvh_bugfix = lambda save_dir, prefix, out_poses, pipeline=None: captured.__setitem__("poses", np.array(out_poses))
    pt.pose2bvh = pt.wavlm_init = pt.load_metadata = None
    sys.path[:0] = [bdir, REF + "/BEAT-TWH-main", REF + "/BEAT-TWH-main/model"]
    import sample as S
    S.mydevice = torch.device("cpu")
    S.batch_size = 1
    mean = np.load("../process/gesture_BEAT_mean_v0.npy")
    std = np.load("../process/gesture_BEAT_std_v0.npy")
    m = mean.shape[-1]
    seed_raw = (mean + std * 0.5 * np.random.RandomState(4711).randn(C.BEAT.n_seed + 2, m)).astype(np.float64)

    class _NP:                      # numpy as the reference module sees it: only the dataset clip is synthetic
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def load(path, *a, **k):
            if "BEAT_dataset" in str(path):
                return seed_raw.copy()
            return np.load(path, *a, **k)
    S.np = _NP()
    from utils.model_util import create_gaussian_diffusion
    from model.mdm import MDM
    diff = create_gaussian_diffusion()
    out = {"wseed": WSEED, "noise_seed": 123456, "skip_timesteps": 997, "real_n_frames": 300, "seed_raw": seed_raw}
    for name, cfg, mode in (("DiffuseStyleGesture+", C.BEAT, 'cross_local_attention4_style1_sample'),
                            ("DiffuseStyleGesture++", C.BEATPP, 'cross_local_attention5_style1_sample'),
                            ("DiffuseStyleGesture", C.BEAT3, 'cross_local_attention3_style1_sample')):
        model = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode=mode, arch='trans_enc', latent_dim=cfg.latent_dim,
                    n_seed=cfg.n_seed, ff_size=cfg.ff_size, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                    style_dim=cfg.style_dim_in, source_audio_dim=cfg.audio_src_dim, audio_feat_dim_latent=cfg.audio_dim)
        missing, unexpected = model.load_state_dict(_to_torch_sd(synth_state_dict(cfg, WSEED)), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        model.eval()
        # textaudio: 300 frames of per-frame features (the stride-long windows of the BEAT set, concatenated)
        ta = np.concatenate([synth_window_inputs(C.BEAT, 1, window=w)["audio"][0] for w in range(3)])[:300]
        args = _ED(n_poses=cfg.n_poses, n_seed=cfg.n_seed, audio_feature_dim=cfg.audio_src_dim, version="v0", name=name,
                   njoints=cfg.njoints)
        style = np.array([1.0, 0.0])
        with NoiseInjector(123456, stream=0) as inj:
            S.inference(args, "/tmp", "g11", torch.from_numpy(ta), diff.p_sample_loop, model, n_frames=0, smoothing=True,
                        skip_timesteps=997, style=style, seed=123456, dataset='BEAT')
            draws = inj.draw
        out[name] = captured["poses"].astype(np.float64)
        out["draws"] = draws
        print("G11", name, captured["poses"].shape, draws, float(np.abs(captured["poses"]).mean()))
    np.savez_compressed(os.path.join(HERE, "g11_clip_dsgplus.npz"), **out)


def gen_dsgpp():
    """G10: DiffuseStyleGesture++ (cond_mode cross_local_attention5_style1: y['seed_last'] through embed_text_last,
    BEAT-TWH-main/model/mdm.py:85-89, :226-264) at the tiny5 dims: conditional and `uncond` forward."""
    sys.path[:0] = [REF + "/BEAT-TWH-main", REF + "/BEAT-TWH-main/model"]
    from model.mdm import MDM
    cfg, ts, B = C.TINY5, 500, 2
    m = MDM(modeltype='', njoints=cfg.njoints, nfeats=1, cond_mode='cross_local_attention5_style1_sample',
            arch='trans_enc', latent_dim=cfg.latent_dim, n_seed=cfg.n_seed, ff_size=cfg.ff_size,
            num_layers=cfg.num_layers, num_heads=cfg.num_heads, style_dim=cfg.style_dim_in,
            source_audio_dim=cfg.audio_src_dim, audio_feat_dim_latent=cfg.audio_dim)
    missing, unexpected = m.load_state_dict(_to_torch_sd(synth_state_dict(cfg, WSEED)), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m.eval()
    y = synth_window_inputs(cfg, B, window=3, seed_pose_scale=0.1)
    x = np.random.RandomState(31 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    g = {"wseed": WSEED, "meta": np.array([B, 0.1, 31 + B, ts], dtype=np.float64)}
    g["tiny5_out"] = m(torch.from_numpy(x), torch.tensor([ts] * B), y=_y_torch(y)).numpy()
    yu = _y_torch(y); yu["uncond"] = True
    g["tiny5_uncond"] = m(torch.from_numpy(x), torch.tensor([ts] * B), y=yu).numpy()
    print("G10", g["tiny5_out"].shape, float(np.abs(g["tiny5_out"]).mean()))
    np.savez_compressed(os.path.join(HERE, "g10_forward_dsgpp.npz"), **g)


def gen_wavlm():
    """G9: the reference's WavLM (mydiffusion_zeggs/WavLM) instantiated with two SMALL configurations that exercise both
    architecture branches -- "large-like" (layer-norm conv extractor, pre-norm encoder, gated relative position bias:
    the WavLM-Large topology the reference loads) and "base-like" (group-norm extractor, post-norm encoder) -- randomly
    initialised by the reference's own init under a fixed torch seed.  Stored: the state dict (the checkpoint format the
    build must ingest), a seeded waveform, `extract_features(wav)[0]` and the wav2wavlm interpolation to 88 frames
    (sample.py:44-48)."""
    sys.path[:0] = [os.path.join(REF, "main/mydiffusion_zeggs/WavLM")]
    from WavLM import WavLM, WavLMConfig
    import torch.nn.functional as F
    cfgs = {
        "large_like": dict(extractor_mode="layer_norm", encoder_layers=3, encoder_embed_dim=64, encoder_ffn_embed_dim=128,
                           encoder_attention_heads=4, layer_norm_first=True, normalize=True, conv_bias=False,
                           conv_feature_layers="[(32,10,5)] + [(32,3,2)] * 4 + [(32,2,2)] * 2", conv_pos=16, conv_pos_groups=4,
                           relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True),
        "base_like": dict(extractor_mode="default", encoder_layers=2, encoder_embed_dim=48, encoder_ffn_embed_dim=96,
                          encoder_attention_heads=4, layer_norm_first=False, normalize=False, conv_bias=False,
                          conv_feature_layers="[(24,10,5)] + [(24,3,2)] * 4 + [(24,2,2)] * 2", conv_pos=16, conv_pos_groups=4,
                          relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True),
    }
    out = {}
    for name, c in cfgs.items():
        torch.manual_seed(1234)
        m = WavLM(WavLMConfig(c)).eval()
        # make the parameters the init leaves at trivial values non-trivial (biases 0, norms 1, grep_a 1)
        g = torch.Generator().manual_seed(99)
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point and (k.endswith("bias") or "layer_norm" in k or "grep_a" in k or k.endswith(".2.1.weight") or k.endswith(".2.weight")):
                v.add_(0.05 * torch.randn(v.shape, generator=g))
        wav = torch.from_numpy(np.random.RandomState(7).randn(2, 16000 * 2 + 321).astype(np.float32) * 0.1)   # 2 windows, ~2 s
        feat = m.extract_features(wav)[0]
        rep = F.interpolate(feat.transpose(1, 2), size=88, align_corners=True, mode="linear").transpose(1, 2)
        out[name + "/cfg"] = np.array(repr(c))
        out[name + "/feat"] = feat.numpy()
        out[name + "/rep88"] = rep.numpy()
        for k, v in m.state_dict().items():
            out[name + "/sd/" + k] = v.numpy()
        print(name, feat.shape, rep.shape, sum(v.numel() for v in m.state_dict().values()), "params")
    out["wav_seed"] = np.array(7)
    np.savez_compressed(os.path.join(HERE, "g9_wavlm_small.npz"), **out)


def gen_wavlm_large():
    """G16: the reference's WavLM at the REAL WavLM-Large topology (24 layers x 1024, 16 heads, ffn 4096: 315.5 M parameters --
    `wavlm.WAVLM_LARGE`), loaded (strict) with the seeded synthetic checkpoint `synth_wavlm_state_dict(WAVLM_LARGE, 5)` -- the
    trained WavLM-Large.pt is not available offline and the weights do not fit a fixture, so both sides regenerate them from the
    seed.  Input: two ZEGGS windows of audio (88 frames x 800 samples = 4.4 s each, sample.py:214-249).  Stored: `extract_features`
    of window 0 and the wav2wavlm interpolation to 88 frames of both (sample.py:44-48)."""
    sys.path[:0] = [os.path.join(REF, "main/mydiffusion_zeggs/WavLM")]
    from WavLM import WavLM, WavLMConfig
    import torch.nn.functional as F
    from diffusestylegesture_amd.synth import synth_wavlm_state_dict
    from diffusestylegesture_amd.wavlm import WAVLM_LARGE
    m = WavLM(WavLMConfig(WAVLM_LARGE)).eval()
    sd = synth_wavlm_state_dict(WAVLM_LARGE, 5)
    r = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    print(r, sum(v.size for v in sd.values()), "params")
    wav = torch.from_numpy(np.random.RandomState(11).randn(2, 88 * 800).astype(np.float32) * 0.1)
    with torch.no_grad():
        feat = m.extract_features(wav)[0]
        rep = F.interpolate(feat.transpose(1, 2), size=88, align_corners=True, mode="linear").transpose(1, 2)
    print("G16", feat.shape, rep.shape, float(feat.std()))
    np.savez_compressed(os.path.join(HERE, "g16_wavlm_large.npz"), wseed=np.array(5), wav_seed=np.array(11),
                        feat0=feat[0].numpy(), rep88=rep.numpy())


if __name__ == "__main__":
    which = sys.argv[1]
    {"zeggs": gen_zeggs, "dsgplus": gen_dsgplus, "clip": gen_clip, "bvh": gen_bvh, "wavlm": gen_wavlm, "wavlm_large": gen_wavlm_large, "dsgpp": gen_dsgpp,
     "clip1000": lambda: gen_clip(0, "g12_clip1000_zeggs.npz", f32=True),
     "bvh1000": lambda: gen_bvh("g12_clip1000_zeggs.npz", "g13_bvh1000_zeggs.npz", full=False),
     "hooks": gen_fn_hooks, "attn3beat": gen_attn3_beat, "dsgplus_caller": gen_dsgplus_caller, "remaining_dims": gen_remaining_dims}[which]()
