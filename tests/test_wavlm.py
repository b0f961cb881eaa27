"""f2: the PyTorch WavLM feature extractor vs `extract_features` of the IMPORTED reference WavLM, for a
WavLM-Large-like topology (layer-norm conv extractor, pre-norm encoder, gated relative position bias) and a Base-like
one (group-norm extractor, post-norm encoder).  Fixture: tests/golden/g9_wavlm_small.npz (make_goldens.py wavlm).
Round 4: the REAL WavLM-Large topology (24 x 1024, 315.5 M parameters, seeded synthetic checkpoint regenerated on both
sides) on two 4.4 s ZEGGS windows -- tests/golden/g16_wavlm_large.npz (make_goldens.py wavlm_large)."""
import ast
import os

import numpy as np
import pytest
import torch

from diffusestylegesture_amd.wavlm import WavLMFeatures, relative_position_buckets, wav2wavlm, _conv_spec
from tests.util import rel_l2


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "g9_wavlm_small.npz"))
    cfg = ast.literal_eval(str(z[name + "/cfg"]))
    sd = {k[len(name + "/sd/"):]: z[k] for k in z.files if k.startswith(name + "/sd/")}
    wav = (np.random.RandomState(int(z["wav_seed"])).randn(2, 16000 * 2 + 321).astype(np.float32) * 0.1)
    return cfg, sd, wav, z[name + "/feat"], z[name + "/rep88"]


@pytest.mark.parametrize("name", ["large_like", "base_like"])
def test_features_match_reference(golden_dir, name):
    cfg, sd, wav, feat, rep88 = _load(golden_dir, name)
    m = WavLMFeatures(cfg, sd)
    out = m.extract_features(torch.from_numpy(wav))[0].numpy()
    assert out.shape == feat.shape
    assert rel_l2(out, feat) < 1e-5                         # fp32 vs fp32, different GEMM grouping only
    # the one-window call of the reference's wav2wavlm, and the batched per-clip cache, give the same frames
    one = wav2wavlm(m, torch.from_numpy(wav[:1])).numpy()
    assert rel_l2(one, rep88[:1]) < 1e-5
    clip = m.clip_features([wav[0], wav[1]]).numpy()
    assert clip.shape == (2, 88, feat.shape[-1])
    assert rel_l2(clip, rep88) < 1e-5


def test_bf16_encoder_is_close(golden_dir):
    cfg, sd, wav, feat, _ = _load(golden_dir, "large_like")
    out = WavLMFeatures(cfg, sd, compute_dtype=torch.bfloat16).extract_features(torch.from_numpy(wav))[0].numpy()
    assert rel_l2(out, feat) < 3e-2


def test_bucket_table_known_answers():
    b = relative_position_buckets(4, 1000, 320, 800)
    assert b.shape == (4, 1000)
    assert int(b[0, 0]) == 0 and int(b[0, 1]) == 161 and int(b[1, 0]) == 1            # sign bit = 160, exact region
    assert int(b[0, 79]) == 160 + 79 and int(b[0, 80]) == 160 + 80                      # first logarithmic bucket
    assert int(b[0, 999]) == 160 + 159                                                  # clamped beyond max_distance
    assert _conv_spec("[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2") == [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["large_like", "base_like"])
def test_features_match_reference_on_gpu(golden_dir, name):
    cfg, sd, wav, feat, rep88 = _load(golden_dir, name)
    m = WavLMFeatures(cfg, sd, device="cuda:0")
    out = m.extract_features(torch.from_numpy(wav))[0].cpu().numpy()
    assert rel_l2(out, feat) < 1e-4
    assert rel_l2(m.clip_features([wav[0], wav[1]]).cpu().numpy(), rep88) < 1e-4


def _load_large(golden_dir):
    from diffusestylegesture_amd.synth import synth_wavlm_state_dict
    from diffusestylegesture_amd.wavlm import WAVLM_LARGE
    z = np.load(os.path.join(golden_dir, "g16_wavlm_large.npz"))
    sd = synth_wavlm_state_dict(WAVLM_LARGE, int(z["wseed"]))
    wav = np.random.RandomState(int(z["wav_seed"])).randn(2, 88 * 800).astype(np.float32) * 0.1
    return WAVLM_LARGE, sd, wav, z["feat0"], z["rep88"]


def test_state_shapes_are_the_checkpoint_contract(golden_dir):
    """wavlm_state_shapes restates the reference module tree's state dict (make_goldens.py loads the synthetic checkpoint built from
    it with strict=True); the small goldens carry real reference state dicts: same names, same shapes."""
    from diffusestylegesture_amd.wavlm import wavlm_state_shapes
    for name in ("large_like", "base_like"):
        cfg, sd, *_ = _load(golden_dir, name)
        want = {k: tuple(v.shape) for k, v in sd.items()}
        assert wavlm_state_shapes(cfg) == want, name


def test_features_match_reference_wavlm_large_topology(golden_dir):
    """The whole 24-layer, 1024-wide encoder on the CPU (fp32): ~10 s of synthesis + ~10 s of forward."""
    cfg, sd, wav, feat0, rep88 = _load_large(golden_dir)
    m = WavLMFeatures(cfg, sd)
    out = m.extract_features(torch.from_numpy(wav[:1]))[0].numpy()
    assert out.shape == (1,) + feat0.shape
    assert rel_l2(out[0], feat0) < 2e-5


@pytest.mark.gpu
def test_features_match_reference_on_gpu_wavlm_large(golden_dir):
    """f2 at real size on MI355X (PyTorch-ROCm, as north_star keeps it): fp32 against the imported reference, bf16 encoder GEMMs
    within the bf16 tolerance; the per-clip cache (all windows in one batched forward) gives the same frames as window by window."""
    import time
    cfg, sd, wav, feat0, rep88 = _load_large(golden_dir)
    m = WavLMFeatures(cfg, sd, device="cuda:0")
    out = m.extract_features(torch.from_numpy(wav))[0].cpu().numpy()
    assert rel_l2(out[0], feat0) < 1e-4
    clip = m.clip_features([wav[0], wav[1]])
    assert rel_l2(clip.cpu().numpy(), rep88) < 1e-4
    one = wav2wavlm(m, torch.from_numpy(wav[1:2])).cpu().numpy()
    assert rel_l2(one, rep88[1:2]) < 1e-4
    m.clip_features([wav[0], wav[1], wav[0], wav[1]])          # warm-up at the timed shape (MIOpen / hipBLASLt heuristics)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.clip_features([wav[0], wav[1], wav[0], wav[1]])
    torch.cuda.synchronize()
    ms32 = (time.perf_counter() - t0) / 3 * 1e3
    mb = WavLMFeatures(cfg, sd, device="cuda:0", compute_dtype=torch.bfloat16)
    e16 = rel_l2(mb.clip_features([wav[0], wav[1]]).cpu().numpy(), rep88)
    mb.clip_features([wav[0], wav[1], wav[0], wav[1]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        mb.clip_features([wav[0], wav[1], wav[0], wav[1]])
    torch.cuda.synchronize()
    ms16 = (time.perf_counter() - t0) / 3 * 1e3
    print(f"WavLM-Large topology, 4 windows x 4.4 s in one batched forward: fp32 {ms32:.1f} ms, bf16 GEMMs {ms16:.1f} ms (rel-L2 {e16:.2e})")
    assert e16 < 3e-2
