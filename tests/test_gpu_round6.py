"""GPU parity tests added in round 6 (all through the C ABI / ctypes shim): the ROWS kernel set, the automatic choice of the set, the noise
transform, the lazy generic progressive loop."""
import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

TOL_FWD = {"fp32": 2e-5, "bf16": 1.2e-2}
TOL_CHAIN = {"fp32": 1e-4, "bf16": 2e-2}


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from diffusestylegesture_amd import lib as L
    return L.default_library()


def _model(cfg, prec, max_batch=1, wseed=20240, **kw):
    from diffusestylegesture_amd.model import DSGDenoiser
    m = DSGDenoiser(cfg, precision=prec, max_batch=max_batch, device=0, **kw)
    m.load_state_dict(synth_state_dict(cfg, wseed))
    return m


def test_rows_kernel_set_vs_oracle(gpu):
    """DSG_KSET_ROWS (ABI 330) at the ZEGGS widths: k_clip_attn + k_ffn on ONE 16-row tile per workgroup (out_proj + LayerNorm1 as its prologue,
    linear1 + GELU + linear2 + residual + LayerNorm2) between BLOCK's pose embedding / local attention and the 16 x 16 pose head -- forward rows
    at batch 12 / 16 / 23 / 46 (256 row tiles: the largest single-lane batch `auto` gives it) and a 30-step DDPM chain at batch 16 against the
    oracle; the rows of a clip are the same bits whatever batch they ride in; `auto` picks it from 1000 token rows."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    small = _model(cfg, "bf16", max_batch=2).set_kernel_set("rows")
    for B in (12, 16, 23, 46):
        y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
        x = np.random.RandomState(100 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 41 + 7) % 1000
        m = _model(cfg, "bf16", max_batch=B)
        assert m.recommend_kernel_set(B, 1) == "rows"
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "rows"                  # what `auto` ran
        for b in sorted({0, B // 2, B - 1}):
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
            assert e < TOL_FWD["bf16"], (B, b, e)
        ys = {k: (v[5:7] if v.shape[0] == B else v) for k, v in y.items()}
        assert np.array_equal(out[5:7], np.asarray(small(x[5:7], ts[5:7], ys))), B
        if B == 16:
            d = create_gaussian_diffusion()
            shape = (B, cfg.njoints, 1, cfg.n_poses)
            got = np.asarray(d.manual_seed(13, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=970))
            assert m.last_sample_path() == "aql" and m.last_kernel_set() == "rows" and m.last_sample_fence_free()
            b = 9
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: sampler.philox.normal_bj1t(shape, 13, k, 1)[b:b + 1], {"y": yb}, skip_timesteps=970)
            assert rel_l2(got[b:b + 1], w) < TOL_CHAIN["bf16"]
            d50 = create_gaussian_diffusion("ddim50")          # config[2]'s arrangement: 50-step DDIM, batch 16 in lock step
            got = np.asarray(d50.manual_seed(14, 2).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}))
            w = sampler.ddim_sample_loop(OracleDiffusion(timestep_respacing="ddim50"), ref, (1,) + shape[1:],
                                         lambda k: sampler.philox.normal_bj1t(shape, 14, k, 2)[b:b + 1], {"y": yb})
            assert m.last_kernel_set() == "rows" and rel_l2(got[b:b + 1], w) < TOL_CHAIN["bf16"]


def test_rows_kernel_set_on_lanes(gpu):
    """4 lanes x 8 clips (what `auto` gives ROWS with several lanes: 1500 token rows over all lanes, the lanes' row tiles within one round of the
    CUs): fence-free AQL packets on 4 queues, a lane reproduces itself alone bit for bit."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip, generate_clips_streams
    cfg, NL, B, K = C.ZEGGS, 4, 8, 2
    m = _model(cfg, "bf16", max_batch=B)
    lanes = [m] + [m.clone() for _ in range(NL - 1)]
    assert m.recommend_kernel_set(B, NL) == "rows" and m.recommend_kernel_set(4, 4) == "rows" and m.recommend_kernel_set(4, 2) == "block" and m.recommend_kernel_set(16, 4) == "stream"
    d = create_gaussian_diffusion()
    feats = [[torch.from_numpy(synth_window_inputs(cfg, B, window=w, clip0=ln * B)["audio"]).cuda() for w in range(K)] for ln in range(NL)]
    got = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=5, skip_timesteps=960, stream_ids=[0, 1, 2, 3])
    assert all(ln.last_kernel_set() == "rows" and ln.last_sample_path() == "aql" and ln.last_sample_fence_free() for ln in lanes) and np.isfinite(got).all()
    lanes[1].set_kernel_set("rows")
    alone = generate_clip(lanes[1], d, feats[1], [1, 0, 0, 0, 0, 0], seed=5, skip_timesteps=960, stream_id=1)
    assert np.array_equal(alone, got[B:2 * B])


def test_noise_stream_vs_oracle(gpu):
    """The round-6 Box-Muller of the noise stream (v_log / v_sqrt + polynomial sincospi; tools/noise_probe.cpp measures the pieces over all 2^24
    arguments) against the float64 transform of the oracle: every element of a [4, 1141, 1, 88] draw within 2e-6 -- as close as the libm form of
    rounds 1-5 (8.3e-7 max over 2^26 calls against 7.5e-7 now, profiles/r06_l_noise_probe.log)."""
    import ctypes as C_
    from oracle import philox
    cfg = C.ZEGGS
    B, J, T = 4, cfg.njoints, cfg.n_poses
    out = np.zeros((B, J, 1, T), np.float32)
    gpu.check(gpu.cdll.dsg_noise(out.ctypes.data, B, J, T, C_.c_uint64(77), C_.c_uint64(5), 9, None))
    want = philox.normal_bj1t((B, J, 1, T), 77, 9, 5)
    assert np.max(np.abs(out - want)) < 2e-6 and abs(float(out.std()) - 1.0) < 5e-3


def test_generic_progressive_loop_is_lazy_and_owns_its_draws(gpu):
    """Round-5 advisor: the generator forms on the GENERIC path (a wrapped model / hooks) reserve their draw indices when they are CREATED and run
    ONE step per next(): a second generator created before the first is consumed draws other noise, and each reproduces the one-call loop."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    cfg = C.TINY
    m = _model(cfg, "fp32", max_batch=1)
    calls = []

    class Wrapped:       # not a DSGDenoiser -> generic loop
        def __call__(self, xx, tt, y=None):
            calls.append(int(tt[0]))
            return m(xx, tt, y)

        def parameters(self):
            return m.parameters()
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    y = {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, 1, window=0, seed_pose_scale=0.3).items()}
    d = create_gaussian_diffusion().manual_seed(3, 1)
    g1 = d.p_sample_loop_progressive(Wrapped(), shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=996)
    g2 = d.p_sample_loop_progressive(Wrapped(), shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=996)
    assert calls == []                                   # nothing has run yet
    first = next(g1)["sample"].clone()
    assert len(calls) == 1                               # ONE denoiser evaluation per next()
    a = [first] + [s["sample"] for s in g1]
    b = [s["sample"] for s in g2]
    assert len(a) == len(b) == 4 and not torch.equal(a[-1], b[-1])
    d.manual_seed(3, 1)
    one = d.p_sample_loop(Wrapped(), shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=996)
    two = d.p_sample_loop(Wrapped(), shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=996)
    assert torch.equal(a[-1], one) and torch.equal(b[-1], two)


def test_rows_kernel_set_bf16w2_vs_oracle(gpu):
    """bf16w2 in the ROWS set (round 6; round-5 verdict item 5) at the ZEGGS widths: k_clip_attn + k_ffn with two-register weight fragments and
    hi + lo A operands -- forward rows at batch 16 / 23 within the mode's 1e-3 of the fp32 oracle (bf16: 1.2e-2), the rows of a clip the same
    bits whatever batch they ride in, a 30-step DDPM chain and the 50-step DDIM of config[2] at batch 16 against the oracle chain; `auto`
    picks ROWS from 800 token rows in one lane; fused guidance stays on TILE."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    small = _model(cfg, "bf16w2", max_batch=2).set_kernel_set("rows")
    for B in (16, 23):
        y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
        x = np.random.RandomState(100 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 41 + 7) % 1000
        m = _model(cfg, "bf16w2", max_batch=B)
        assert m.recommend_kernel_set(B, 1) == "rows" and m.recommend_kernel_set(3, 1) == "tile"
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "rows"                  # what `auto` ran
        for b in sorted({0, B // 2, B - 1}):
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
            assert e < 1e-3, (B, b, e)
        ys = {k: (v[5:7] if v.shape[0] == B else v) for k, v in y.items()}
        assert np.array_equal(out[5:7], np.asarray(small(x[5:7], ts[5:7], ys))), B
        if B == 16:
            d = create_gaussian_diffusion()
            shape = (B, cfg.njoints, 1, cfg.n_poses)
            got = np.asarray(d.manual_seed(13, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=970))
            assert m.last_sample_path() == "aql" and m.last_kernel_set() == "rows" and m.last_sample_fence_free()
            b = 9
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: sampler.philox.normal_bj1t(shape, 13, k, 1)[b:b + 1], {"y": yb}, skip_timesteps=970)
            assert rel_l2(got[b:b + 1], w) < 1.5e-3
            d50 = create_gaussian_diffusion("ddim50")          # config[2]'s arrangement: 50-step DDIM, batch 16 in lock step
            got = np.asarray(d50.manual_seed(14, 2).ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}))
            w = sampler.ddim_sample_loop(OracleDiffusion(timestep_respacing="ddim50"), ref, (1,) + shape[1:],
                                         lambda k: sampler.philox.normal_bj1t(shape, 14, k, 2)[b:b + 1], {"y": yb})
            assert m.last_kernel_set() == "rows" and rel_l2(got[b:b + 1], w) < 1.5e-3


def test_local_attention_one_wave_form_is_bit_identical(gpu, monkeypatch):
    """Round 6: k_loc as ONE wave per (head, window, clip) (what 32+ clips per lane run: 2048 items; DSG_LOC64_FROM is the threshold's test hook) against
    the 256-thread form at the ZEGGS and BEAT dims: the same bits, both within tolerance of the oracle -- with a key mask and with mask_local=None."""
    from oracle.mdm import MDMOracle
    for cfg in (C.ZEGGS, C.CONFIGS["beat"]):
        sd = synth_state_dict(cfg, 20240)
        ref = MDMOracle(sd, cfg)
        B = 3
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.2)
        ym = dict(y)
        mask = np.ones_like(np.asarray(y["mask_local"]))
        mask[..., 5:9] = 0
        ym["mask_local"] = mask.astype(np.asarray(y["mask_local"]).dtype)
        yn = dict(y)
        yn["mask_local"] = None
        x = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = [10, 500, 999]
        for prec in ("fp32", "bf16"):
            outs = {}
            for frm in ("1", "1000000"):
                monkeypatch.setenv("DSG_LOC64_FROM", frm)
                m = _model(cfg, prec, max_batch=B).set_kernel_set("tile")
                outs[frm] = [np.asarray(m(x, ts, yy)) for yy in (y, ym, yn)]
                for o, yy in zip(outs[frm], (y, ym, yn)):
                    assert rel_l2(o, ref(x, ts, yy)) < TOL_FWD[prec], (cfg.name, prec, frm)
            assert all(np.array_equal(p, q) for p, q in zip(outs["1"], outs["1000000"])), (cfg.name, prec)


@pytest.mark.parametrize("cfg", [C.BEAT, C.TWH], ids=lambda c: c.name)
def test_rows_kernel_set_at_dsgplus_widths(gpu, cfg):
    """Round 6 (round-5 verdict item 7): ROWS at latent_dim 384 / 512 -- the streamed pose embedding (K over two workgroups), k_clip_attn_w (the attention half per (clip, head), the
    clip's rows through the LDS in chunks) + k_ffn<OP> on one 16-row tile per workgroup (at 512 W_o leads the weight ring instead of waiting in registers), the streaming pose head:
    no QKV GEMM, no k_attn_op_w, no ff-split, no slabs.  Forward rows at batch 16 (what `auto` picks from 9 clips)
    against the oracle, a clip's rows the same bits at batch 2, a 20-step DDPM chain; BLOCK stays the choice at 8 clips and under fused guidance."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    B = 16
    y = synth_window_inputs(cfg, B, window=1, clip0=2, seed_pose_scale=0.2)
    x = np.random.RandomState(7).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 53 + 11) % 1000
    m = _model(cfg, "bf16", max_batch=B)
    assert [m.recommend_kernel_set(b, 1) for b in (4, 8, 9, 13, 16)] == ["block", "block", "rows" if cfg.latent_dim == 384 else "block", "rows", "rows"]
    assert [m.recommend_kernel_set(b, 4) for b in (2, 3, 4, 8)] == ["block", "block", "rows", "rows"]
    out = np.asarray(m(x, ts, y))
    assert m.last_kernel_set() == "rows"
    for b in (0, 7, B - 1):
        yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
        e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
        assert e < TOL_FWD["bf16"], (cfg.name, b, e)
    small = _model(cfg, "bf16", max_batch=2).set_kernel_set("rows")
    ys = {k: (v[5:7] if v.shape[0] == B else v) for k, v in y.items()}
    assert np.array_equal(out[5:7], np.asarray(small(x[5:7], ts[5:7], ys)))
    d = create_gaussian_diffusion()
    shape = (B, cfg.njoints, 1, cfg.n_poses)
    got = np.asarray(d.manual_seed(21, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=980))
    assert m.last_sample_path() == "aql" and m.last_kernel_set() == "rows"
    b = 11
    yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
    w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: sampler.philox.normal_bj1t(shape, 21, k, 3)[b:b + 1], {"y": yb}, skip_timesteps=980)
    assert rel_l2(got[b:b + 1], w) < TOL_CHAIN["bf16"]
    with pytest.raises(NotImplementedError):
        _model(cfg, "fp32", max_batch=2).set_kernel_set("rows")
    # batch 8 and 32 (round-5 verdict item 7: "oracle rows at batch 8 / 32"): the streamed pose embedding with K over two workgroups (k_ws2<PARTIAL, 17 / 18, 2>)
    # on 8 / 38 full 32-row blocks and a partial one; at 32 clips the 302 row tiles need two rounds of the CUs
    for Bn in (8, 32):
        yn = synth_window_inputs(cfg, Bn, window=2, clip0=1, seed_pose_scale=0.2)
        xn = np.random.RandomState(Bn).randn(Bn, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        tn = (np.arange(Bn) * 31 + 5) % 1000
        mn = _model(cfg, "bf16", max_batch=Bn).set_kernel_set("rows")
        on = np.asarray(mn(xn, tn, yn))
        assert mn.last_kernel_set() == "rows"
        for b in (0, Bn - 1):
            yb = {k: (v[b:b + 1] if v.shape[0] == Bn else v) for k, v in yn.items()}
            e = rel_l2(on[b:b + 1], ref(xn[b:b + 1], [int(tn[b])], yb))
            assert e < TOL_FWD["bf16"], (cfg.name, Bn, b, e)
        del mn
    # four lanes x 8 clips: from 1200 rows the QKV projection is the weight-stationary streaming GEMM (k_ws at K = 384 / 512; at 512 only with several lanes), and with
    # >= 3 lanes whose row tiles together exceed one round of the CUs (4 x 76 = 304) k_ffn<OP> runs on 32-row blocks (W_o leading the weight ring at both widths, at 512
    # the fp32 LayerNorm1 rows through X1) -- bit-identical to the forms a lane runs alone
    B2, NL = 8, 4
    m8 = _model(cfg, "bf16", max_batch=B2).set_kernel_set("rows")
    lanes = [m8] + [m8.clone() for _ in range(NL - 1)]
    shape2 = (B2, cfg.njoints, 1, cfg.n_poses)
    ys = [{"y": synth_window_inputs(cfg, B2, window=w, clip0=B2 * w, seed_pose_scale=0.2)} for w in range(NL)]
    multi = d.manual_seed(9, 0).p_sample_loop_multi(lanes, shape2, ys, seeds=[9] * NL, stream_ids=list(range(NL)), skip_timesteps=990)
    for i in range(NL):
        alone = d.manual_seed(9, i).p_sample_loop(lanes[i], shape2, clip_denoised=False, model_kwargs=ys[i], skip_timesteps=990)
        assert lanes[i].last_kernel_set() == "rows" and np.array_equal(np.asarray(multi[i]), np.asarray(alone)), (cfg.name, i)


def test_rows_kernel_set_at_beat_v2_pose_width(gpu):
    """BEAT "v2" (motion_dim 1141 -> 1152 padded pose features at latent_dim 384): the ROWS set with the pose embedding streamed WHOLE (k_ws2<PARTIAL, 18>: K = 1152 fits one
    workgroup, no K split) and a 9-panel streaming pose head next to k_clip_attn_w / k_ffn<OP> -- forward rows at batch 16 against the oracle."""
    from oracle.mdm import MDMOracle
    cfg = C.BEATV2
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    B = 16
    y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
    x = np.random.RandomState(11).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 61 + 3) % 1000
    m = _model(cfg, "bf16", max_batch=B)
    assert m.recommend_kernel_set(B, 1) == "rows"
    out = np.asarray(m(x, ts, y))
    assert m.last_kernel_set() == "rows"
    for b in (0, 9, B - 1):
        yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
        e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
        assert e < TOL_FWD["bf16"], (b, e)
