"""f2: the PyTorch WavLM feature extractor vs `extract_features` of the IMPORTED reference WavLM, for a
WavLM-Large-like topology (layer-norm conv extractor, pre-norm encoder, gated relative position bias) and a Base-like
one (group-norm extractor, post-norm encoder).  Fixture: tests/golden/g9_wavlm_small.npz (make_goldens.py wavlm)."""
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
