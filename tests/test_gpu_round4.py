"""MI355X parity tests (-m gpu), round 4: every kernel set at the DSG+ dims (BEAT / TWH / BEAT++; batch 1, 8, 32) against the oracle;
the DSG+ window loop on sampling lanes (bench.py --config beat --clips-per-gpu 16); the state-dict contract errors on the device
library (round 3 only ran them under the emulator); the STREAM set with more than one row block per persistent workgroup at
latent_dim 128 (round-3 advisor); the per-step forms of the progressive generators.
Tolerances as in test_gpu_round3.py (bf16 <= 2x the values measured on MI355X)."""
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


def _model(cfg, prec, max_batch=1, wseed=20240):
    from diffusestylegesture_amd.model import DSGDenoiser
    m = DSGDenoiser(cfg, precision=prec, max_batch=max_batch, device=0)
    m.load_state_dict(synth_state_dict(cfg, wseed))
    return m


def test_state_dict_contract_errors_on_the_device_library(gpu):
    """load_model_wo_clip's contract (main/utils/model_util.py:8-12) through libdsg_hip.so itself: unexpected key, missing key at
    finalize, size mismatch, the tolerated clip_model.* keys -- and a handle that failed to load still loads a good dict."""
    from diffusestylegesture_amd.model import DSGDenoiser
    cfg = C.TINY
    sd = synth_state_dict(cfg, 1)
    m = DSGDenoiser(cfg, precision="bf16", max_batch=1, device=0)
    with pytest.raises(ValueError, match="unexpected key"):
        m.load_state_dict({"bogus.weight": np.zeros(3, np.float32)})
    sd2 = dict(sd)
    sd2.pop("input_process2.bias")
    with pytest.raises(ValueError, match="missing key"):
        m.load_state_dict(sd2)
    with pytest.raises(ValueError, match="size mismatch"):
        m.load_state_dict({"input_process2.bias": np.zeros(3, np.float32)})
    m.load_state_dict(dict(sd, **{"clip_model.x": np.zeros(2, np.float32)}))
    y = synth_window_inputs(cfg, 1, window=1)
    x = np.random.RandomState(3).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    assert np.isfinite(np.asarray(m(x, np.array([10]), y))).all()


@pytest.mark.parametrize("cfg", [C.BEAT, C.TWH, C.BEATPP], ids=lambda c: c.name)
def test_every_kernel_set_at_dsgplus_dims_batch_1_8_32(gpu, cfg):
    """Round-3 verdict item 1: each kernel set at the DSG+ dims at batch 1 (the sets config[4] can run: auto, latency, tile), batch 8
    and batch 32 (tile / block; 4832 token rows) -- forward rows against the oracle in bf16 (fp32 at batch 1 and 8), a 12-step chain
    at batch 8.  K = 384 / 512 GEMMs run their whole k range as one fragment batch since round 4."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    for B, sets, precs in ((1, ("auto", "latency", "tile"), ("fp32", "bf16")), (8, ("tile", "block", "rows"), ("fp32", "bf16")), (32, ("tile", "block", "rows"), ("bf16",))):
        y = synth_window_inputs(cfg, B, window=2, clip0=5, seed_pose_scale=0.2)
        x = np.random.RandomState(B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 31 + 7) % 1000
        rows = sorted({0, B // 2, B - 1})
        want = {b: ref(x[b:b + 1], [int(ts[b])], {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}) for b in rows}
        for prec in precs:
            m = _model(cfg, prec, max_batch=B)
            for ks in sets:
                if ks == "rows" and (prec != "bf16" or cfg.latent_dim not in (384, 512)):      # (round 6: ROWS at the DSG+ widths is a bf16 set; 32 clips: two rounds of row tiles)
                    continue
                m.set_kernel_set(ks)
                out = np.asarray(m(x, ts, y))
                if ks != "auto":
                    assert m.last_kernel_set() == ks
                else:
                    assert m.last_kernel_set() == ("tile" if cfg.latent_dim >= 384 else "latency")
                for b in rows:
                    e = rel_l2(out[b:b + 1], want[b])
                    assert e < TOL_FWD[prec], (cfg.name, B, prec, ks, b, e)
            if B == 8 and prec == "bf16":
                d = create_gaussian_diffusion()
                m.set_kernel_set("block")
                shape = (B, cfg.njoints, 1, cfg.n_poses)
                got = np.asarray(d.manual_seed(11, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=988))
                b = 3
                yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
                nf = lambda k: sampler.philox.normal_bj1t(shape, 11, k, 2)[b:b + 1]
                w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], nf, {"y": yb}, skip_timesteps=988)
                assert rel_l2(got[b:b + 1], w) < TOL_CHAIN[prec], (cfg.name, rel_l2(got[b:b + 1], w))


@pytest.mark.parametrize("cfg", [C.BEAT, C.TWHPP], ids=lambda c: c.name)
def test_dsgplus_clip_loop_on_lanes_vs_single_lane_and_oracle(gpu, cfg):
    """generate_clips_streams_dsgplus: 4 lanes x batch 2 at DSG+ dims, 3 windows x 40 steps, AQL packets on 4 queues; every lane ==
    generate_clip_dsgplus on that lane alone (bit for bit, same kernel set), and one clip against the oracle's DSG+ clip driver."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip_dsgplus, generate_clips_streams_dsgplus
    NL, B, K, n_run, frames = 4, 2, 3, 40, 300
    skip = 1000 - n_run
    m = _model(cfg, "bf16", max_batch=B)
    lanes = [m] + [m.clone() for _ in range(NL - 1)]
    d = create_gaussian_diffusion()
    ins = [[synth_window_inputs(cfg, B, window=w, clips=[2 * ln, 2 * ln + 1], seed_pose_scale=0.2) for w in range(K)] for ln in range(NL)]
    # (DSG++: the caller hands stride-long feature windows and drops the last n_seed frames itself; synth gives T - 2S frames)
    pad = (lambda a: np.concatenate([a, a[:, :cfg.n_seed]], 1)) if cfg.variant == 5 else (lambda a: a)
    feats = [[torch.from_numpy(pad(y["audio"])).cuda() for y in il] for il in ins]
    seed0s = [torch.from_numpy(il[0]["seed"]).cuda() for il in ins]
    lasts = [torch.from_numpy(il[0]["seed_last"]).cuda() for il in ins] if cfg.variant == 5 else None
    style = [1] + [0] * (cfg.style_dim_in - 1)
    ks = m.recommend_kernel_set(B, NL)
    got = generate_clips_streams_dsgplus(lanes, d, feats, style, seed0s, frames, seed=21, skip_timesteps=skip, stream_ids=[3, 4, 5, 6], seed_lasts=lasts)
    assert got.shape == (NL * B, frames, cfg.njoints // 3) and np.isfinite(got).all()
    assert all(ln.last_kernel_set() == ks and ln.last_sample_path() == "aql" for ln in lanes)
    assert all(ln.kernel_set() == "auto" for ln in lanes)
    for ln in (0, 3):
        lanes[ln].set_kernel_set(ks)
        want = generate_clip_dsgplus(lanes[ln], d, feats[ln], style, seed0s[ln], frames, seed=21, skip_timesteps=skip, stream_id=[3, 4, 5, 6][ln],
                                     seed_last=None if lasts is None else lasts[ln])
        assert np.array_equal(got[ln * B:(ln + 1) * B], want), ln
    # ... and the oracle's DSG+ clip driver (round-4 verdict 8a: the docstring promised it): clip 7 = batch element 1 of lane 3, every
    # window's chain on the noise rows of that element (draw index of window c, step k: c (1 + n_run) + k of stream 6)
    from oracle import philox, sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    ln, b = 3, 1
    ref, od = MDMOracle(synth_state_dict(cfg, 20240), cfg), OracleDiffusion()
    shape = (B, cfg.njoints, 1, cfg.n_poses)

    def sample_window(c, yy):
        nf = lambda k: philox.normal_bj1t(shape, 21, c * (1 + n_run) + k, 6)[b:b + 1]
        return sampler.p_sample_loop(od, ref, (1,) + shape[1:], nf, {"y": yy}, skip_timesteps=skip)
    feats_np = [pad(il["audio"])[b:b + 1] for il in ins[ln]]
    want = sampler.dsgplus_clip(sample_window, cfg, feats_np, style, ins[ln][0]["seed"][b:b + 1], frames,
                                seed_last=ins[ln][0]["seed_last"][b:b + 1] if cfg.variant == 5 else None)
    e = rel_l2(got[ln * B + b], want)
    print(f"{cfg.name}: lane {ln} clip {b} of the 4 x 2 DSG+ call vs the oracle clip driver: rel-L2 {e:.2e}")
    assert e < TOL_CHAIN["bf16"], e


def test_stream_set_several_blocks_per_workgroup_at_latent_128(gpu):
    """Round-3 advisor (medium): k_ws<EPI, 8> staged V^T / pose-head tiles past its 16 KB activation buffer once a persistent
    workgroup owned a second row block (tiny dims, batch >= 468; `auto` picks STREAM there).  On the device: the first, middle and
    last clips of a batch of 480 equal the same clips sampled four at a time under the same set, bit for bit, and match the oracle."""
    from oracle.mdm import MDMOracle
    cfg, B = C.TINY, 480
    sd = synth_state_dict(cfg, 20240)
    yb = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    xb = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 2 + 3) % 1000
    big = _model(cfg, "bf16", max_batch=B)
    out = np.asarray(big(xb, ts, yb))
    assert big.last_kernel_set() == "stream" and np.isfinite(out).all()
    small = _model(cfg, "bf16", max_batch=4).set_kernel_set("stream")
    ref = MDMOracle(sd, cfg)
    for lo in (0, 236, B - 4):
        ys = {k: (v[lo:lo + 4] if v.shape[0] == B else v) for k, v in yb.items()}
        want = np.asarray(small(xb[lo:lo + 4], ts[lo:lo + 4], ys))
        assert np.array_equal(out[lo:lo + 4], want), lo
        assert rel_l2(out[lo:lo + 4], ref(xb[lo:lo + 4], list(ts[lo:lo + 4]), ys)) < 2.5e-2


def test_handle_created_after_a_destroyed_one_is_clean(gpu):
    """Round-4 finding (tools/debug_rowdep*.py): uncached device memory that went back to the HIP allocator and came back for another
    buffer was not reliably coherent -- a batch-200 handle created after a batch-170 handle had been destroyed computed every frame
    row >= 4096 wrong (up to 100 % off, all kernel sets, DSG_UC=0 clean).  Uncached blocks now live in a process-wide pool and are
    reused for uncached requests only.  Sizes 170 -> 200 and 170 -> 180 are the ones that failed."""
    import gc
    cfg = C.TINY
    sd = synth_state_dict(cfg, 20240)

    def inputs(B):
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
        x = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        return x, (np.arange(B) * 2 + 3) % 1000, y
    small = _model(cfg, "bf16", max_batch=4).set_kernel_set("block")
    for pre, B in ((170, 200), (170, 180), (200, 170)):
        x, t, y = inputs(pre)
        p = _model(cfg, "bf16", max_batch=pre).set_kernel_set("block")
        p(x, t, y)
        del p
        gc.collect()
        x, t, y = inputs(B)
        big = _model(cfg, "bf16", max_batch=B).set_kernel_set("block")
        out = np.asarray(big(x, t, y))
        for lo in (0, B // 2, B - 4):
            ys = {k: (v[lo:lo + 4] if v.shape[0] == B else v) for k, v in y.items()}
            assert np.array_equal(out[lo:lo + 4], np.asarray(small(x[lo:lo + 4], t[lo:lo + 4], ys))), (pre, B, lo)
        del big
        gc.collect()


def test_rows_do_not_depend_on_the_batch(gpu):
    """Within a kernel set a clip's rows are the same bits whatever batch they ride in -- ZEGGS, every set, batch 8 (712 rows: the
    3-waves-per-SIMD LayerNorm GEMMs of TILE) and batch 48 (4272 rows) against batch 2.  (Round 4 found with this test that k_attn_op2 was one ulp
    off k_attn_op on the device -- a differently contracted LayerNorm; k_attn_op2 is retired in round 6, experiments/dsg_rejected_kernels.h.)"""
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    B = 48
    y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    x = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 7 + 3) % 1000
    for ks, Bs in (("tile", (8,)), ("block", (8, 48)), ("stream", (8, 48)), ("rows", (8, 48))):
        small = _model(cfg, "bf16", max_batch=2).set_kernel_set(ks)
        for Bb in Bs:
            big = _model(cfg, "bf16", max_batch=Bb).set_kernel_set(ks)
            out = np.asarray(big({8: x[:8], 48: x}[Bb], ts[:Bb], {k: (v[:Bb] if v.shape[0] == B else v) for k, v in y.items()}))
            for lo in (0, Bb - 2):
                ys = {k: (v[lo:lo + 2] if v.shape[0] == B else v) for k, v in y.items()}
                assert np.array_equal(out[lo:lo + 2], np.asarray(small(x[lo:lo + 2], ts[lo:lo + 2], ys))), (ks, Bb, lo)


def test_stream_set_with_fused_ffn_vs_oracle_and_lanes(gpu):
    """Round 4: the STREAM set runs the feed-forward half of a layer as ONE kernel (k_ffn: linear1 + GELU + linear2 + residual +
    LayerNorm2, the hidden activations never leave LDS; the next QKV and the pose head are direct streaming GEMMs, no k_ln_frag):
    forward rows at batch 3 / 16 / 40 and a 30-step chain at batch 16 against the oracle; 4 lanes x batch 16 (the arrangement of
    bench.py --clips-per-gpu 64) fence-free on 4 queues, a lane reproduces itself alone bit for bit; the recommendation thresholds."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip, generate_clips_streams
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    for B in (3, 16, 40):
        y = synth_window_inputs(cfg, B, window=2, clip0=1, seed_pose_scale=0.2)
        x = np.random.RandomState(B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 23 + 5) % 1000
        m = _model(cfg, "bf16", max_batch=B).set_kernel_set("stream")
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "stream"
        for b in sorted({0, B // 2, B - 1}):
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
            assert e < TOL_FWD["bf16"], (B, b, e)
        if B == 16:
            d = create_gaussian_diffusion()
            shape = (B, cfg.njoints, 1, cfg.n_poses)
            got = np.asarray(d.manual_seed(11, 2).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=970))
            assert m.last_sample_path() == "aql" and m.last_sample_fence_free()
            b = 7
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: sampler.philox.normal_bj1t(shape, 11, k, 2)[b:b + 1], {"y": yb}, skip_timesteps=970)
            assert rel_l2(got[b:b + 1], w) < TOL_CHAIN["bf16"]
    NL, B, K = 4, 16, 2
    m = _model(cfg, "bf16", max_batch=B)
    lanes = [m] + [m.clone() for _ in range(NL - 1)]
    assert m.recommend_kernel_set(16, 4) == "stream" and m.recommend_kernel_set(12, 4) == "rows" and m.recommend_kernel_set(4, 4) == "rows" and m.recommend_kernel_set(4, 2) == "block"      # (round 6: ROWS in between, from 1000 rows over all lanes)
    assert m.recommend_kernel_set(16, 1) == "rows" and m.recommend_kernel_set(48, 1) == "stream"
    d = create_gaussian_diffusion()
    feats = [[torch.from_numpy(synth_window_inputs(cfg, B, window=w, clip0=ln * B)["audio"]).cuda() for w in range(K)] for ln in range(NL)]
    got = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=5, skip_timesteps=960, stream_ids=[0, 1, 2, 3])
    assert all(ln.last_kernel_set() == "stream" and ln.last_sample_path() == "aql" for ln in lanes) and np.isfinite(got).all()
    lanes[1].set_kernel_set("stream")
    alone = generate_clip(lanes[1], d, feats[1], [1, 0, 0, 0, 0, 0], seed=5, skip_timesteps=960, stream_id=1)
    assert np.array_equal(alone, got[B:2 * B])


def test_block_set_with_the_ffn_split_over_hidden_vs_oracle(gpu, monkeypatch):
    """Round 4: below the STREAM threshold the BLOCK set runs the feed-forward half of a layer as k_ffn_part (4 workgroups per
    32-row block, each a quarter of ff: partial linear2 slabs in fp32) + k_ffn_ln (slab sum in a fixed order + bias + residual +
    LayerNorm2), and the next QKV projection / the pose head as direct GEMMs.  Forward rows at batch 1 / 16 / 23 (1 x 23 needs a
    second round of k_ffn_part workgroups) and a 30-step chain at batch 16 against the oracle; the round-3 kernels
    (DSG_FFN_SPLIT=0: linear1, linear2, LayerNorm-on-read) agree to bf16 noise; a row does not depend on the batch."""
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    for B in (1, 16, 23):
        y = synth_window_inputs(cfg, B, window=1, clip0=3, seed_pose_scale=0.2)
        x = np.random.RandomState(100 + B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = (np.arange(B) * 41 + 7) % 1000
        m = _model(cfg, "bf16", max_batch=B).set_kernel_set("block")
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "block"
        for b in sorted({0, B // 2, B - 1}):
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            e = rel_l2(out[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb))
            assert e < TOL_FWD["bf16"], (B, b, e)
        if B == 16:
            monkeypatch.setenv("DSG_FFN_SPLIT", "0")          # (read once, at dsg_create: a handle created under the switch)
            old = np.asarray(_model(cfg, "bf16", max_batch=B).set_kernel_set("block")(x, ts, y))
            monkeypatch.delenv("DSG_FFN_SPLIT")
            assert 0 < rel_l2(out, old) < TOL_FWD["bf16"]            # a different set of kernels ran, same function
            monkeypatch.setenv("DSG_CLIP_ATTN", "0")         # round 5: QKV GEMM + k_attn_op instead of k_clip_attn + the out_proj / LayerNorm1 prologue
            old5 = np.asarray(_model(cfg, "bf16", max_batch=B).set_kernel_set("block")(x, ts, y))
            monkeypatch.delenv("DSG_CLIP_ATTN")
            assert 0 < rel_l2(out, old5) < TOL_FWD["bf16"] and 0 < rel_l2(old, old5)
            for b in (0, B - 1):
                yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
                assert rel_l2(old5[b:b + 1], ref(x[b:b + 1], [int(ts[b])], yb)) < TOL_FWD["bf16"]
            small = _model(cfg, "bf16", max_batch=2).set_kernel_set("block")
            ys = {k: (v[5:7] if v.shape[0] == B else v) for k, v in y.items()}
            assert np.array_equal(out[5:7], np.asarray(small(x[5:7], ts[5:7], ys)))
            d = create_gaussian_diffusion()
            shape = (B, cfg.njoints, 1, cfg.n_poses)
            got = np.asarray(d.manual_seed(13, 1).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=970))
            assert m.last_sample_path() == "aql" and m.last_kernel_set() == "block"
            b = 9
            yb = {k: (v[b:b + 1] if v.shape[0] == B else v) for k, v in y.items()}
            w = sampler.p_sample_loop(OracleDiffusion(), ref, (1,) + shape[1:], lambda k: sampler.philox.normal_bj1t(shape, 13, k, 1)[b:b + 1], {"y": yb}, skip_timesteps=970)
            assert rel_l2(got[b:b + 1], w) < TOL_CHAIN["bf16"]


def test_ffn_64_row_blocks_with_four_large_lanes_bit_identical(gpu, monkeypatch):
    """Round 4: with 4 lanes of >= 4000 token rows each (bench.py --clips-per-gpu 192 / 256) k_ffn runs on 64-row blocks -- half the
    weight bytes per row through the CUs' load paths.  Same waves, same k order: a lane's clips are bit-identical to the lane run
    alone (32-row blocks), and so is a forward with the block shape forced (DSG_FFN_RT4)."""
    import torch
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.sample import generate_clip, generate_clips_streams
    cfg = C.ZEGGS
    NL, B = 4, 48
    m = _model(cfg, "bf16", max_batch=B)
    y = synth_window_inputs(cfg, B, window=1, clip0=2, seed_pose_scale=0.2)
    x = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = (np.arange(B) * 19 + 3) % 1000
    monkeypatch.setenv("DSG_FFN_RT4", "0")              # (read once, at dsg_create: handles created under the switch)
    a = np.asarray(_model(cfg, "bf16", max_batch=B).set_kernel_set("stream")(x, ts, y))
    monkeypatch.setenv("DSG_FFN_RT4", "1")
    b = np.asarray(_model(cfg, "bf16", max_batch=B).set_kernel_set("stream")(x, ts, y))
    monkeypatch.delenv("DSG_FFN_RT4")                   # (32-row blocks: a 32-fragment weight ring; 64-row blocks: 12 fragments)
    assert np.array_equal(a, b) and np.isfinite(a).all()
    m.set_kernel_set("auto")
    lanes = [m] + [m.clone() for _ in range(NL - 1)]
    d = create_gaussian_diffusion()
    feats = [[torch.from_numpy(synth_window_inputs(cfg, B, window=0, clip0=ln * B)["audio"]).cuda()] for ln in range(NL)]
    got = generate_clips_streams(lanes, d, feats, [1, 0, 0, 0, 0, 0], seed=9, skip_timesteps=985, stream_ids=[0, 1, 2, 3])
    assert all(ln.last_kernel_set() == "stream" and ln.last_sample_path() == "aql" for ln in lanes) and np.isfinite(got).all()
    lanes[2].set_kernel_set("stream")
    alone = generate_clip(lanes[2], d, feats[2], [1, 0, 0, 0, 0, 0], seed=9, skip_timesteps=985, stream_id=2)
    assert np.array_equal(alone, got[2 * B:3 * B])
