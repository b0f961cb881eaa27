"""CPU tests of the round-6 additions under the SIMT emulator (the product sources, unchanged, compiled for the host): the ROWS kernel set
(k_ffn on one 16-row tile per workgroup behind k_clip_attn), the noise transform, the automatic choice of the set."""
import numpy as np
import pytest

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
from tests.util import rel_l2


def test_rows_kernel_set_vs_oracle_and_batch_independence(emu_lib):
    """DSG_KSET_ROWS (ABI 330) at the tiny dims: forward rows at batch 3 and 5 (an odd number of row tiles) and a 6-step DDPM chain against the
    fp32 oracle; a row's bits do not depend on the batch it rides in; a lane over the same weights reproduces the handle; fp32 handles
    refuse the set like STREAM (bf16w2 has it: the next test)."""
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.TINY
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    small = DSGDenoiser(cfg, precision="bf16", max_batch=2, library=emu_lib).set_kernel_set("rows")
    small.load_state_dict(sd)
    for B in (3, 5):
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
        x = np.random.RandomState(B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = [10, 500, 999, 3, 77][:B]
        m = DSGDenoiser(cfg, precision="bf16", max_batch=B, library=emu_lib).set_kernel_set("rows")
        m.load_state_dict(sd)
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "rows" and rel_l2(out, ref(x, ts, y)) < 1.2e-2
        ys = {k: (v[B - 2:B] if v.shape[0] == B else v) for k, v in y.items()}
        assert np.array_equal(out[B - 2:B], np.asarray(small(x[B - 2:B], ts[B - 2:B], ys)))
    shape = (3, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 3, window=1, seed_pose_scale=0.3)
    m = DSGDenoiser(cfg, precision="bf16", max_batch=3, library=emu_lib).set_kernel_set("rows")
    m.load_state_dict(sd)
    d = create_gaussian_diffusion(library=emu_lib)
    got = np.asarray(d.manual_seed(11, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=994))
    want = sampler.p_sample_loop(OracleDiffusion(), ref, shape, sampler.philox_noise_fn(shape, 11, 3), {"y": y}, skip_timesteps=994)
    assert m.last_kernel_set() == "rows" and rel_l2(got, want) < 2e-2
    lane = m.clone()
    assert lane.kernel_set() == "rows"
    again = np.asarray(d.manual_seed(11, 3).p_sample_loop(lane, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=994))
    assert np.array_equal(got, again)
    with pytest.raises(NotImplementedError):
        DSGDenoiser(cfg, precision="fp32", max_batch=2, library=emu_lib).set_kernel_set("rows")


def test_rows_kernel_set_bf16w2_vs_oracle(emu_lib):
    """bf16w2 in the ROWS set (round 6; round-5 verdict item 5): k_clip_attn + k_ffn with the two-register weight fragments and hi + lo A operands
    (embedding output, attention rows, LayerNorm1 / LayerNorm2 rows, `hidden`) at the tiny dims -- forward rows at batch 3 and 5 and a 6-step DDPM
    chain within the mode's 1e-3 of the fp32 oracle (bf16: 1.2e-2); a row's bits do not depend on the batch; STREAM / BLOCK stay refused; the
    automatic choice is ROWS from 800 token rows in one lane (1400 over several lanes of at least 300) and TILE below / under fused guidance."""
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.TINY
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    small = DSGDenoiser(cfg, precision="bf16w2", max_batch=2, library=emu_lib).set_kernel_set("rows")
    small.load_state_dict(sd)
    for B in (3, 5):
        y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
        x = np.random.RandomState(B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
        ts = [10, 500, 999, 3, 77][:B]
        m = DSGDenoiser(cfg, precision="bf16w2", max_batch=B, library=emu_lib).set_kernel_set("rows")
        m.load_state_dict(sd)
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == "rows" and rel_l2(out, ref(x, ts, y)) < 1e-3
        ys = {k: (v[B - 2:B] if v.shape[0] == B else v) for k, v in y.items()}
        assert np.array_equal(out[B - 2:B], np.asarray(small(x[B - 2:B], ts[B - 2:B], ys)))
    shape = (3, cfg.njoints, 1, cfg.n_poses)
    y = synth_window_inputs(cfg, 3, window=1, seed_pose_scale=0.3)
    m = DSGDenoiser(cfg, precision="bf16w2", max_batch=3, library=emu_lib).set_kernel_set("rows")
    m.load_state_dict(sd)
    d = create_gaussian_diffusion(library=emu_lib)
    got = np.asarray(d.manual_seed(11, 3).p_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=994))
    want = sampler.p_sample_loop(OracleDiffusion(), ref, shape, sampler.philox_noise_fn(shape, 11, 3), {"y": y}, skip_timesteps=994)
    assert m.last_kernel_set() == "rows" and rel_l2(got, want) < 1.5e-3
    for kset in ("stream", "block"):
        with pytest.raises(NotImplementedError):
            DSGDenoiser(cfg, precision="bf16w2", max_batch=2, library=emu_lib).set_kernel_set(kset)
    assert [m.recommend_kernel_set(b, 1) for b in (1, 34, 35, 200)] == ["tile", "tile", "rows", "rows"]
    assert [m.recommend_kernel_set(b, 4) for b in (13, 14, 15, 16)] == ["tile", "tile", "tile", "rows"]


def test_auto_kernel_set_table_round6(emu_lib):
    """What DSG_KSET_AUTO resolves to (dsg_recommend_kernel_set; tiny dims: 23 token rows per clip): ROWS from 800 token rows in one lane while the
    row tiles fit the 256 CUs in one round, STREAM beyond; with several lanes ROWS from 1000 rows over all lanes and 250 per lane, STREAM once the
    lanes' row tiles exceed 300; BLOCK from 500 rows in one lane / 250 per lane below that."""
    m = DSGDenoiser(C.TINY, precision="bf16", max_batch=2, library=emu_lib)
    m.load_state_dict(synth_state_dict(C.TINY, 20240))
    r = m.recommend_kernel_set
    assert [r(b, 1) for b in (1, 2, 3, 21, 22, 34, 35, 178, 179)] == ["latency", "latency", "tile", "tile", "block", "block", "rows", "rows", "stream"]
    assert [r(b, 4) for b in (1, 2, 10, 11, 14, 17, 52, 53)] == ["latency", "tile", "tile", "rows", "rows", "rows", "rows", "stream"]
    assert [r(b, 2) for b in (10, 11, 21, 22)] == ["tile", "block", "block", "rows"]
    f = DSGDenoiser(C.TINY, precision="fp32", max_batch=2, library=emu_lib)
    f.load_state_dict(synth_state_dict(C.TINY, 20240))
    assert [f.recommend_kernel_set(b, 1) for b in (1, 3, 43, 44, 200)] == ["tile", "tile", "tile", "block", "block"]


def test_noise_stream_vs_oracle(emu_lib):
    """The round-6 Box-Muller (v_log / v_sqrt + polynomial sincospi on the device; libm stand-ins under the emulator) against the float64 transform
    of the oracle: every element of a [2, J, 1, T] draw within 2e-6."""
    import ctypes as C_
    from oracle import philox
    cfg = C.TINY
    B, J, T = 2, cfg.njoints, cfg.n_poses
    out = np.zeros((B, J, 1, T), np.float32)
    emu_lib.check(emu_lib.cdll.dsg_noise(out.ctypes.data, B, J, T, C_.c_uint64(77), C_.c_uint64(5), 9, None))
    want = philox.normal_bj1t((B, J, 1, T), 77, 9, 5)
    assert np.max(np.abs(out - want)) < 2e-6


def test_tile_gemm_xcd_balanced_mapping_from_16_row_tiles(emu_lib):
    """Round 6: from 16 row tiles the column groups past the last full round of 8 are dealt out by row tile over the padded grid columns (gemm_body).
    Tiny dims: every K = D GEMM has such groups (QKV 6, pose head 2) -- batch 12 (18 row tiles) on the TILE and ROWS sets against the oracle, and a
    clip's rows are the bits they are at batch 2 (5 row tiles: the fixed map)."""
    from oracle.mdm import MDMOracle
    cfg = C.TINY
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    B = 12
    y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    x = np.random.RandomState(B).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = [(37 * b + 5) % 1000 for b in range(B)]
    for prec, kset, tol in (("bf16", "tile", 1.2e-2), ("fp32", "tile", 2e-5), ("bf16", "rows", 1.2e-2), ("bf16w2", "rows", 1e-3)):
        m = DSGDenoiser(cfg, precision=prec, max_batch=B, library=emu_lib).set_kernel_set(kset)
        m.load_state_dict(sd)
        out = np.asarray(m(x, ts, y))
        assert m.last_kernel_set() == kset and rel_l2(out, ref(x, ts, y)) < tol, (prec, kset)
        small = DSGDenoiser(cfg, precision=prec, max_batch=2, library=emu_lib).set_kernel_set(kset)
        small.load_state_dict(sd)
        ys = {k: (v[7:9] if v.shape[0] == B else v) for k, v in y.items()}
        assert np.array_equal(out[7:9], np.asarray(small(x[7:9], ts[7:9], ys))), (prec, kset)


def test_local_attention_one_wave_form_is_bit_identical(emu_lib, monkeypatch):
    """Round 6: k_loc's scores / P V on fp32 matrix instructions by one wave; from 2048 (head, window, clip) items the kernel runs as ONE wave per item
    (DSG_LOC64_FROM is the test hook for the threshold; two waves per item in the shipped form).  Both forms against the oracle (incl. a key mask and mask_local=None), bit-identical to each other."""
    from oracle.mdm import MDMOracle
    cfg = C.TINY
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    B = 3
    y = synth_window_inputs(cfg, B, window=1, seed_pose_scale=0.3)
    ym = dict(y)
    mask = np.ones_like(np.asarray(y["mask_local"]))
    mask[..., 3:6] = 0
    ym["mask_local"] = mask.astype(np.asarray(y["mask_local"]).dtype)
    yn = dict(y)
    yn["mask_local"] = None
    x = np.random.RandomState(5).randn(B, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    ts = [10, 500, 999]
    for prec, tol in (("fp32", 2e-5), ("bf16", 1.2e-2)):
        outs = {}
        for frm in ("1", "1000000"):
            monkeypatch.setenv("DSG_LOC64_FROM", frm)
            m = DSGDenoiser(cfg, precision=prec, max_batch=B, library=emu_lib).set_kernel_set("tile")
            m.load_state_dict(sd)
            outs[frm] = [np.asarray(m(x, ts, yy)) for yy in (y, ym, yn)]
            for o, yy in zip(outs[frm], (y, ym, yn)):
                assert rel_l2(o, ref(x, ts, yy)) < tol, (prec, frm)
        assert all(np.array_equal(p, q) for p, q in zip(outs["1"], outs["1000000"])), prec


@pytest.mark.parametrize("name", ["beat", "twh"])
def test_rows_kernel_set_at_dsgplus_widths(emu_lib, name):
    """Round 6: ROWS at latent_dim 384 / 512 (streamed pose embedding with K over two workgroups, k_clip_attn_w -- one pass over the rows at 384, two at 512 --, k_ffn<OP> on 16-row
    tiles -- at 512 W_o leads the weight ring --, the streaming pose head k_ws<OUT, 24 / 32>) -- one forward at batch 1 against
    the oracle under the emulator (the GPU test has batch 16, batch independence and a chain)."""
    from oracle.mdm import MDMOracle
    cfg = C.CONFIGS[name]
    sd = synth_state_dict(cfg, 20240)
    ref = MDMOracle(sd, cfg)
    y = synth_window_inputs(cfg, 1, window=1, seed_pose_scale=0.2)
    x = np.random.RandomState(3).randn(1, cfg.njoints, 1, cfg.n_poses).astype(np.float32)
    m = DSGDenoiser(cfg, precision="bf16", max_batch=1, library=emu_lib).set_kernel_set("rows")
    m.load_state_dict(sd)
    out = np.asarray(m(x, [417], y))
    assert m.last_kernel_set() == "rows" and rel_l2(out, ref(x, [417], y)) < 1.2e-2
    assert [m.recommend_kernel_set(b, 1) for b in (8, 9, 13, 27, 28, 48)] == ["block", "rows" if cfg.latent_dim == 384 else "block", "rows", "rows", "rows", "rows"]      # (past one round of the CUs too: 1 x 32 clips 702 vs 844 us BLOCK)
