"""MI355X parity tests (-m gpu), round 3: BASELINE config[3]'s real per-GPU arrangement (4 sampling lanes x batch 4) against
the oracle row by row, with the kernel set that ran asserted; the fence-free loop soaked in that arrangement over the full
4 windows x 1000 steps; a 1000-step chain at config[4]'s dims (TWH, latent 512); kernel sets as explicit, sticky properties
of a lane; the guidance wrapper's fall-back to the generic loop; one real RCCL rank through bench.py.
Tolerances (rel-L2 on normalised poses): fp32 kernels 2e-5 per forward / 1e-4 per chain; bf16 kernels 1.2e-2 per forward,
2e-2 per chain (<= 2x the values measured on MI355X: 4.5e-3 .. 6.7e-3 per forward, 9.3e-3 after 1000 steps)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from tests.conftest import ROOT
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

TOL_FWD = {"fp32": 2e-5, "bf16": 1.2e-2}       # as in test_gpu_parity.py: bf16 <= 2x the measured 4.5e-3 .. 6.7e-3
TOL_CHAIN = {"fp32": 1e-4, "bf16": 2e-2}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from diffusestylegesture_amd import lib as L
    return L.default_library()


def _model(cfg, prec, max_batch=1, wseed=20240):
    from diffusestylegesture_amd.model import DSGDenoiser
    m = DSGDenoiser(cfg, precision=prec, max_batch=max_batch, device=0)
    m.load_state_dict(synth_state_dict(cfg, wseed))
    return m


# ---------------------------------------------------------------------------------------------------------------------
# config[3]: "128 ZEGGS clips sharded one-clip-per-stream" = 16 clips per GPU = 4 lanes (own HSA queue each) x batch 4,
# the arrangement bench.py --clips-per-gpu 16 runs.  Every clip has its own conditioning and its own Philox rows; the
# oracle samples a clip on its own (rows and lanes are independent), through the same window loop / stitching.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,NL,B,kset", [("fp32", 4, 4, "block"), ("bf16", 4, 4, "rows"), ("bf16", 4, 16, "stream")])
def test_config3_arrangement_4_lanes_x_batch_4_vs_oracle(gpu, prec, NL, B, kset):
    """4 lanes x batch 4 = config[3]'s 16 clips per GPU (fp32: BLOCK; bf16: ROWS since round 6 -- 186.8 vs 194.2 us per step); 4 lanes x
    batch 16 = 64 clips per GPU, the arrangement from which the lanes run the STREAM set (round 3)."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clips_streams
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, K, n_run = C.ZEGGS, 2, 60
    skip = 1000 - n_run
    m = _model(cfg, prec, max_batch=B)
    lanes = [m] + [m.clone() for _ in range(NL - 1)]
    d = create_gaussian_diffusion()
    feats_np = [[synth_window_inputs(cfg, B, window=w, clip0=ln * B)["audio"] for w in range(K)] for ln in range(NL)]
    feats = [[torch.from_numpy(f).cuda() for f in fl] for fl in feats_np]
    style = [0, 0, 1, 0, 0, 0]
    sids = [7 + ln for ln in range(NL)]
    got = generate_clips_streams(lanes, d, feats, style, seed=4242, skip_timesteps=skip, stream_ids=sids)
    assert got.shape == (NL * B, K * cfg.stride - cfg.n_seed, cfg.njoints) and np.isfinite(got).all()
    # the arrangement that ships: AQL packets on 4 queues, fence-free, the kernel set recommended for 4 lanes x 356 / 1424 rows
    assert m.recommend_kernel_set(B, NL) == kset
    assert all(ln.last_kernel_set() == kset and ln.last_sample_path() == "aql" and ln.last_sample_fence_free() for ln in lanes)
    ref, od = MDMOracle(synth_state_dict(cfg, 20240), cfg), OracleDiffusion()
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    worst = 0.0
    for ln, b in ((0, 0), (0, B - 1), (1, 1), (2, 2), (3, 0), (3, B - 1)):
        def sample_window(c, y, ln=ln, b=b):
            nf = lambda k: philox.normal_bj1t(shape, 4242, c * (1 + n_run) + k, sids[ln])[b:b + 1]
            return sampler.p_sample_loop(od, ref, (1,) + shape[1:], nf, {"y": y}, skip_timesteps=skip)
        want = sampler.zeggs_clip(sample_window, cfg, [f[b:b + 1] for f in feats_np[ln]], style)
        e = rel_l2(got[ln * B + b], want)
        worst = max(worst, e)
        assert e < TOL_CHAIN[prec], (ln, b, e)
    print(f"{NL} lanes x batch {B} ({kset}) {prec}: worst rel-L2 of 6 clips vs oracle = {worst:.3e}")
    assert not np.array_equal(got[0], got[1]) and not np.array_equal(got[0], got[B])
    # and a lane reproduces itself bit for bit when sampled alone under the same kernel set (round 4: the multi-lane call puts the
    # lanes' own sets back on exit -- `auto` here, which alone would pick another set for batch 4 than for 4 lanes x 4)
    from diffusestylegesture_amd.sample import generate_clip
    assert lanes[2].kernel_set() == "auto"
    lanes[2].set_kernel_set(kset)
    alone = generate_clip(lanes[2], d, feats[2], style, seed=4242, skip_timesteps=skip, stream_id=sids[2])
    assert lanes[2].last_kernel_set() == kset and np.array_equal(alone, got[2 * B:3 * B])


def test_config3_fence_free_soak_full_clip(gpu, monkeypatch):
    """The fence-free loop (uncached loop buffers, AQL packets without acquire / release) in the arrangement that ships -- 4 lanes x
    batch 4 -- over the WHOLE config workload, 4 windows x 1000 steps = 560 000 fence-free packets on 4 queues at once: bit-identical
    to cached buffers + agent-scope fences (DSG_UC=0).  One stale line anywhere in 4000 steps would show."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import DSGDenoiser
    from diffusestylegesture_amd.sample import generate_clips_streams
    cfg, NL, B, K = C.ZEGGS, 4, 4, 4
    feats = [[torch.from_numpy(synth_window_inputs(cfg, B, window=w, clip0=ln * B)["audio"]).cuda() for w in range(K)] for ln in range(NL)]
    outs = {}
    for uc in ("1", "0"):
        monkeypatch.setenv("DSG_UC", uc)
        m = DSGDenoiser(cfg, precision="bf16", max_batch=B)
        m.load_state_dict(synth_state_dict(cfg, 20240))
        lanes = [m] + [m.clone() for _ in range(NL - 1)]
        d = create_gaussian_diffusion()
        outs[uc] = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=99, skip_timesteps=0, stream_ids=[0, 1, 2, 3])
        assert all(ln.last_sample_path() == "aql" and ln.last_kernel_set() == "rows" for ln in lanes)      # (round 6: what 4 lanes x batch 4 run)
        assert all(ln.last_sample_fence_free() == (uc == "1") for ln in lanes)
    assert outs["1"].shape == (16, 312, cfg.njoints) and np.isfinite(outs["1"]).all()
    assert np.array_equal(outs["0"], outs["1"])


def test_stream_set_fence_free_at_large_batch(gpu, monkeypatch):
    """Handles that can run the STREAM set keep uncached loop buffers and fence-free packets at every max_batch (round 3: 1 x 64
    505 -> 479 us/step).  2 lanes x batch 32 (2848 token rows per lane: STREAM by the automatic rule), 2 windows x 300 steps:
    bit-identical to cached buffers + agent-scope fences, and the default really is the fence-free path."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import DSGDenoiser
    from diffusestylegesture_amd.sample import generate_clips_streams
    cfg, NL, B, K = C.ZEGGS, 2, 32, 2
    feats = [[torch.from_numpy(synth_window_inputs(cfg, B, window=w, clip0=ln * B)["audio"]).cuda() for w in range(K)] for ln in range(NL)]
    outs = {}
    for uc in ("default", "0"):
        if uc == "default":
            monkeypatch.delenv("DSG_UC", raising=False)
        else:
            monkeypatch.setenv("DSG_UC", uc)
        m = DSGDenoiser(cfg, precision="bf16", max_batch=B)
        m.load_state_dict(synth_state_dict(cfg, 20240))
        lanes = [m, m.clone()]
        d = create_gaussian_diffusion()
        outs[uc] = generate_clips_streams(lanes, d, feats, [0, 1, 0, 0, 0, 0], seed=7, skip_timesteps=700, stream_ids=[0, 1])
        assert all(ln.last_sample_path() == "aql" and ln.last_kernel_set() == "stream" for ln in lanes)
        assert all(ln.last_sample_fence_free() == (uc == "default") for ln in lanes)
    assert outs["0"].shape == (64, 152, cfg.njoints) and np.isfinite(outs["0"]).all()
    assert np.array_equal(outs["0"], outs["default"])


def test_twh_1000_step_chain_bf16_vs_oracle(gpu):
    """config[4] dims (TWH: latent 512, K = 2232 pose features, 151 tokens; the TILE kernel set): one whole window, 1000 DDPM
    steps in bf16 against the fp32 oracle -- the drift number config[4] (16 windows x 1000 steps) rests on.  ~40 s of CPU."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.TWH
    sd = synth_state_dict(cfg, 20240)
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.1)
    r = sampler.p_sample_loop(OracleDiffusion(), MDMOracle(sd, cfg), shape, sampler.philox_noise_fn(shape, 21, 3), {"y": y})
    errs = {}
    for prec in ("bf16", "fp32"):
        m = _model(cfg, prec)
        s = create_gaussian_diffusion().manual_seed(21, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y})
        assert m.last_kernel_set() == "tile" and m.last_sample_path() == "aql"
        errs[prec] = rel_l2(s, r)
    print(f"TWH 1000-step chain vs oracle: bf16 {errs['bf16']:.3e}, fp32 {errs['fp32']:.3e}")
    assert errs["fp32"] < TOL_CHAIN["fp32"] and errs["bf16"] < TOL_CHAIN["bf16"], errs


def test_kernel_sets_are_sticky_lane_properties(gpu):
    """dsg_set_kernel_set / dsg_last_kernel_set on the hardware: every set against the oracle at ZEGGS dims (batch 3, 40 steps),
    clones inherit the source's set, AUTO follows the batch, and a lane of batch 2 gives the same bits alone and inside
    dsg_sample_multi (round-2 advisor finding: the choice used to depend on the number of lanes in the call)."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, B = C.ZEGGS, 3
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, B, window=2, clip0=5, seed_pose_scale=0.2)
    ref, od = MDMOracle(synth_state_dict(cfg, 20240), cfg), OracleDiffusion()
    y1 = {k: (v[1:2] if k != "mask_local" else v) for k, v in y.items()}
    want = sampler.p_sample_loop(od, ref, (1,) + shape[1:], lambda k: philox.normal_bj1t(shape, 8, k, 2)[1:2], {"y": y1}, skip_timesteps=960)
    m = _model(cfg, "bf16", max_batch=B)
    d = create_gaussian_diffusion()
    res = {}
    for kset in ("auto", "latency", "tile", "block"):
        m.set_kernel_set(kset)
        res[kset] = np.asarray(d.manual_seed(8, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=960)).copy()
        assert m.last_kernel_set() == ("tile" if kset == "auto" else kset)
        assert rel_l2(res[kset][1], want[0]) < TOL_CHAIN["bf16"], kset
    assert np.array_equal(res["auto"], res["tile"]) and not np.array_equal(res["tile"], res["block"])
    c = m.clone()
    assert c.recommend_kernel_set(2, 4) == "tile" and c.recommend_kernel_set(2, 1) == "latency"
    shape2 = (2,) + shape[1:]
    ys = [{"y": synth_window_inputs(cfg, 2, window=w, clip0=2 * w, seed_pose_scale=0.2)} for w in range(2)]
    for kset in ("latency", "tile"):
        lanes = [m.set_kernel_set(kset), c.set_kernel_set(kset)]
        multi = d.manual_seed(9, 0).p_sample_loop_multi(lanes, shape2, ys, seeds=[9, 9], stream_ids=[0, 1], skip_timesteps=950)
        for i in range(2):
            alone = d.manual_seed(9, i).p_sample_loop(lanes[i], shape2, clip_denoised=False, model_kwargs=ys[i], skip_timesteps=950)
            assert lanes[i].last_kernel_set() == kset and np.array_equal(np.asarray(multi[i]), np.asarray(alone)), (kset, i)


@pytest.mark.parametrize("cfg_name", ["beat", "twh", "beatpp"])
def test_block_set_at_dsgplus_dims_batch_8(gpu, cfg_name):
    """The BLOCK kernel set at DSG+ dims (latent 384 / 512, head dim 96 / 128, 151 tokens: 32-row block GEMMs with the 512-wide row
    buffer, V^T through LDS in aligned token groups with heads that straddle a 64-column group) -- batch 8 = 1208 token rows is
    where AUTO selects it; forward with distinct rows and timesteps against the oracle in fp32 and bf16, and a short DDPM chain."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, B = C.CONFIGS[cfg_name], 8
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    x = np.random.RandomState(8).randn(*shape).astype(np.float32)
    ts = (np.arange(B) * 97 + 11) % 1000
    for prec in ("fp32", "bf16"):
        m = _model(cfg, prec, max_batch=B)
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "block"
        for b in (0, 5, B - 1):
            yb = {k: (v[b:b + 1] if k != "mask_local" and v is not None else v) for k, v in y.items()}
            e = rel_l2(out[b], ref(x[b:b + 1], [int(ts[b])], yb)[0])
            assert e < TOL_FWD[prec], (prec, b, e)
    s = np.asarray(create_gaussian_diffusion().manual_seed(4, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=988))
    assert m.last_kernel_set() == "block" and m.last_sample_path() == "aql"
    b = 6
    yb = {k: (v[b:b + 1] if k != "mask_local" and v is not None else v) for k, v in y.items()}
    r = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: philox.normal_bj1t(shape, 4, k, 2)[b:b + 1], {"y": yb}, skip_timesteps=988)
    e = rel_l2(s[b], r[0])
    print(f"{cfg_name} batch 8, BLOCK set, 12-step chain row {b}: rel-L2 {e:.3e}")
    assert e < TOL_CHAIN["bf16"]


@pytest.mark.parametrize("kset", ["block", "stream", "rows"])
def test_guidance_in_the_batched_kernel_sets(gpu, kset):
    """Classifier-free guidance fused into the step loop (cfg_sampler.py:8-31: conditional rows + their unconditional twins in one
    batch, combined in the pose-head epilogue) in the BLOCK and STREAM sets, whose state shadow is fragment-major and whose pose
    embedding streams it: 6 clips + 6 twins = 1068 token rows, 10 DDPM steps, two scales, against the oracle's two evaluations
    per step -- rows 0 and 5."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import ClassifierFreeSampleModel
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, B = C.ZEGGS, 6
    m = _model(cfg, "bf16", max_batch=2 * B).set_kernel_set(kset)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    y = synth_window_inputs(cfg, B, window=1, clip0=2, seed_pose_scale=0.3)
    scale = np.linspace(0.5, 2.5, B).astype(np.float32)
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion().manual_seed(3, 9)
    s = np.asarray(d.p_sample_loop(ClassifierFreeSampleModel(m), shape, clip_denoised=False, model_kwargs={"y": dict(y, scale=scale)}, skip_timesteps=990))
    assert m.last_kernel_set() == kset and m.last_sample_path() == "aql" and np.isfinite(s).all()
    for b in (0, B - 1):
        yb = {k: (v[b:b + 1] if k != "mask_local" else v) for k, v in y.items()}
        yb["scale"] = scale[b:b + 1]
        r = sampler.p_sample_loop(OracleDiffusion(), sampler.CFGModel(ref), (1,) + shape[1:], lambda k, b=b: philox.normal_bj1t(shape, 3, k, 9)[b:b + 1],
                                  {"y": yb}, skip_timesteps=990)
        e = rel_l2(s[b], r[0])
        assert e < 2 * TOL_CHAIN["bf16"], (kset, b, e)


def test_stream_and_block_sets_at_tiny_dims(gpu):
    """The K = 128 instantiations (k_ws<.., 8>, k_ws2<.., 2>, k_ln_frag<2>, k_attn_op<.., 2, 2>) that the ZEGGS tests never launch, on the GPU:
    tiny dims (latent 128, 23 tokens -- batch elements misaligned to the 4-token groups of V^T), batch 23, forward + 10-step chain."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg, B = C.TINY, 23
    sd = synth_state_dict(cfg, 77)
    ref = MDMOracle(sd, cfg)
    y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    x = np.random.RandomState(B).randn(*shape).astype(np.float32)
    ts = np.arange(B) * 40 + 3
    want = ref(x, list(ts), y)
    for kset in ("stream", "block", "rows"):
        m = _model(cfg, "bf16", max_batch=B, wseed=77).set_kernel_set(kset)
        assert rel_l2(np.asarray(m(x, ts, y)), want) < TOL_FWD["bf16"] and m.last_kernel_set() == kset
        s = np.asarray(create_gaussian_diffusion().manual_seed(3, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990))
        r = sampler.p_sample_loop(OracleDiffusion(), ref, shape, lambda k: philox.normal_bj1t(shape, 3, k, 1), {"y": y}, skip_timesteps=990)
        assert m.last_sample_path() == "aql" and rel_l2(s, r) < TOL_CHAIN["bf16"], kset


@pytest.mark.parametrize("B", [3, 16, 48])
def test_stream_kernel_set_vs_oracle(gpu, B):
    """Kernel set "stream" (dsg_stream.h: weight-stationary persistent GEMMs, 32x32x16 MFMA, global->LDS staging, 64-row blocks;
    LayerNorm once per row, V^T through LDS in aligned token groups, two query tiles per attention workgroup from 4000 rows) at
    ZEGGS dims: a forward with ragged last blocks (267 / 1424 / 4272 rows) and a 40-step DDPM chain with distinct rows against the
    oracle, row by row; DDIM on top at batch 16 (config[2]'s shape)."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, B, window=2, clip0=11, seed_pose_scale=0.2)
    ref = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    m = _model(cfg, "bf16", max_batch=B).set_kernel_set("stream")
    x = np.random.RandomState(B).randn(*shape).astype(np.float32)
    ts = (np.arange(B) * 61 + 5) % 1000
    out = np.asarray(m(x, ts, y))
    assert m.last_kernel_set() == "stream"
    rows = [0, B - 1] if B > 3 else [0, 1, 2]
    for b in rows:
        yb = {k: (v[b:b + 1] if k != "mask_local" else v) for k, v in y.items()}
        e = rel_l2(out[b], ref(x[b:b + 1], [int(ts[b])], yb)[0])
        assert e < 1.2e-2, (b, e)
    d = create_gaussian_diffusion()
    s = np.asarray(d.manual_seed(5, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=960))
    assert m.last_kernel_set() == "stream" and m.last_sample_path() == "aql" and np.isfinite(s).all()
    od = OracleDiffusion()
    for b in rows[:2]:
        yb = {k: (v[b:b + 1] if k != "mask_local" else v) for k, v in y.items()}
        r = sampler.p_sample_loop(od, ref, (1,) + shape[1:], lambda k, b=b: philox.normal_bj1t(shape, 5, k, 3)[b:b + 1], {"y": yb}, skip_timesteps=960)
        e = rel_l2(s[b], r[0])
        print(f"stream set, batch {B}, row {b}: 40-step chain rel-L2 {e:.3e}")
        assert e < TOL_CHAIN["bf16"], (b, e)
    if B == 16:
        d50 = create_gaussian_diffusion("ddim50")
        s50 = np.asarray(d50.manual_seed(6, 1).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, eta=0.0))
        yb = {k: (v[7:8] if k != "mask_local" else v) for k, v in y.items()}
        r = sampler.ddim_sample_loop(OracleDiffusion(timestep_respacing="ddim50"), ref, (1,) + shape[1:],
                                     lambda k: philox.normal_bj1t(shape, 6, k, 1)[7:8], {"y": yb}, eta=0.0)
        assert rel_l2(s50[7], r[0]) < TOL_CHAIN["bf16"]


def test_guidance_wrapper_without_room_for_twins_uses_generic_loop(gpu):
    """Round-2 advisor finding: ClassifierFreeSampleModel around a denoiser of max_batch < 2B must still sample (generic loop: two
    library calls per step, the same Philox stream), not raise -- and agree with the fused 2B-row loop."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import ClassifierFreeSampleModel
    cfg = C.ZEGGS
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.3)
    yt = {k: torch.from_numpy(v).cuda() for k, v in y.items()}
    yt["scale"] = torch.tensor([2.5], device="cuda")
    d = create_gaussian_diffusion()
    small, big = _model(cfg, "fp32", max_batch=1), _model(cfg, "fp32", max_batch=2)
    a = d.manual_seed(3, 9).p_sample_loop(ClassifierFreeSampleModel(small), shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=990)
    b = d.manual_seed(3, 9).p_sample_loop(ClassifierFreeSampleModel(big), shape, clip_denoised=False, model_kwargs={"y": yt}, skip_timesteps=990)
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


def test_progressive_generators_match_dump_steps(gpu):
    """p_sample_loop_progressive / ddim_sample_loop_progressive (gaussian_diffusion.py:673-740, :938-1003) as generators of
    {"sample": x_{t-1}} per step: the same intermediate samples as dump_steps, the last one equal to the plain loop's result."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    cfg = C.ZEGGS
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 1, window=0, seed_pose_scale=0.2)
    m = _model(cfg, "fp32")
    d = create_gaussian_diffusion()
    full = np.asarray(d.manual_seed(4, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990))
    steps = [np.asarray(o["sample"]) for o in d.manual_seed(4, 1).p_sample_loop_progressive(m, shape, clip_denoised=False, model_kwargs={"y": y},
                                                                                              skip_timesteps=990)]
    assert len(steps) == 10 and np.array_equal(steps[-1], full)
    dumped = d.manual_seed(4, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=990, dump_steps=[0, 4, 9])
    assert all(np.array_equal(np.asarray(dumped[i]), steps[s]) for i, s in enumerate((0, 4, 9)))
    d5 = create_gaussian_diffusion("ddim5")
    full5 = np.asarray(d5.manual_seed(4, 2).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, eta=0.3))
    steps5 = [np.asarray(o["sample"]) for o in d5.manual_seed(4, 2).ddim_sample_loop_progressive(m, shape, clip_denoised=False,
                                                                                                 model_kwargs={"y": y}, eta=0.3)]
    assert len(steps5) == 5 and rel_l2(steps5[-1], full5) < 1e-6


def test_bench_one_real_rccl_rank(gpu):
    """bench.py under torch.distributed.run with ONE rank and backend "nccl": real init_process_group, real dist.gather /
    all_reduce (RCCL), the per-rank AQL queue created after set_device(LOCAL_RANK) on the agent whose PCI address matches the
    HIP device (printed once per rank) -- what every rank of the 8-GPU run does, on the one GPU a lease has.  The rate must
    agree with the un-launched run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-postprocess", "--sub-records", "off"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True, timeout=600, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ranked = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                             "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist"] + args,
                            capture_output=True, text=True, timeout=600, env=env)
    assert ranked.returncode == 0, ranked.stderr[-2000:]
    a = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    b = json.loads([ln for ln in ranked.stdout.splitlines() if ln.startswith("{")][-1])
    assert b["n_gpus"] == 1 and b["sample_path"] == "aql" and b["fence_free_packets"] and b["collective_backend"] == "nccl"
    assert "HSA agent" in ranked.stderr and "rank-local HIP device 0" in ranked.stderr
    print(f"bench: plain {a['value']:.1f} frames/s, one RCCL rank {b['value']:.1f} frames/s")
    assert abs(b["value"] / a["value"] - 1.0) < 0.05


@pytest.mark.parametrize("cfg", [C.TWH3, C.TWHPP, C.BEATV2], ids=lambda c: c.name)
def test_remaining_name_x_dataset_dims_vs_reference(gpu, golden_dir, cfg):
    """G15: TWH under "DiffuseStyleGesture" (attention3) / "DiffuseStyleGesture++" (attention5) and BEAT "v2" (njoints 1141): forward
    vs the imported reference at those dims (BEAT-TWH-main/mydiffusion_beat_twh/sample.py:297-323 accepts every pair)."""
    g = np.load(os.path.join(golden_dir, "g15_forward_remaining_dims.npz"))
    _, sps, rs, ts = g[cfg.name + "_meta"]
    y = synth_window_inputs(cfg, 1, window=2, seed_pose_scale=float(sps))
    x = np.random.RandomState(int(rs)).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    for prec, tol in (("fp32", 2e-5), ("bf16", 1.2e-2)):
        m = _model(cfg, prec, wseed=int(g["wseed"]))
        e = rel_l2(m(x, np.array([int(ts)]), y), g[cfg.name + "_out"])
        assert e < tol, (cfg.name, prec, e)


def test_dsgplus_command_line_every_name_x_dataset(gpu, tmp_path):
    """sample_plus.main for the pairs round 2 refused: TWH x {DiffuseStyleGesture, ++} and BEAT v2 -- checkpoint file in, poses out."""
    import torch
    from diffusestylegesture_amd import sample_plus
    ms = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "beat_twh_mean_std.npz"))
    rs = np.random.RandomState(4)
    for name, dataset, version, cfg in (("DiffuseStyleGesture", "TWH", "v0", C.TWH3), ("DiffuseStyleGesture++", "TWH", "v0", C.TWHPP),
                                        ("DiffuseStyleGesture+", "BEAT", "v2", C.BEATV2)):
        ck = str(tmp_path / f"{cfg.name}.pt")
        torch.save({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 20240).items()}, ck)
        ta = np.concatenate([synth_window_inputs(C.TWH if dataset == "TWH" else C.BEAT, 1, window=w)["audio"][0] for w in range(2)])[:170]
        np.save(str(tmp_path / "ta.npy"), ta)
        argv = ["--model_path", ck, "--features_npy", str(tmp_path / "ta.npy"), "--name", name, "--dataset", dataset, "--version", version,
                "--save_dir", str(tmp_path / ("o" + cfg.name)), "--skip_timesteps", "997"]
        if version == "v2":
            np.savez(str(tmp_path / "ms.npz"), mean=np.zeros(1141), std=np.ones(1141))
            np.save(str(tmp_path / "seed.npy"), 0.1 * rs.randn(cfg.n_seed, 1141))
            argv += ["--mean_std_npz", str(tmp_path / "ms.npz")]
            width = 1141
        else:
            mean, std = ms[dataset + "_mean"], ms[dataset + "_std"]
            np.save(str(tmp_path / "seed.npy"), mean + std * 0.5 * rs.randn(cfg.n_seed + 2, mean.shape[-1]))
            width = cfg.njoints // 3
        argv += ["--seed_npy", str(tmp_path / "seed.npy")]
        if cfg.variant == 5:
            argv += ["--seed_last_npy", str(tmp_path / "seed.npy")]
        res = np.load(sample_plus.main(argv))
        assert res.shape == (170, width) and np.isfinite(res).all(), (name, dataset, version, res.shape)
