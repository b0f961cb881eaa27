"""Seeded synthetic weights and inputs for the denoiser (no trained checkpoint is available offline).

The generator is stream-stable: every tensor is drawn from its own
`np.random.RandomState((crc32(name) + seed) % 2**32)`, so any subset can be regenerated
anywhere (dev container, GPU box) bit-identically.  Names and shapes are the reference
`state_dict` contract (`main/model/mdm.py:10-151`; `BEAT-TWH-main/model/mdm.py:11-118`;
loader `main/utils/model_util.py:8-12`).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import DSGConfig, VARIANT_DSG


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 32))


def _mat(name, shape, seed, scale=0.5):
    fan_in = shape[-1]
    return (_rs(name, seed).randn(*shape) * (scale / np.sqrt(fan_in))).astype(np.float32)


def _vec(name, n, seed, scale=0.02, offset=0.0):
    return (offset + scale * _rs(name, seed).randn(n)).astype(np.float32)


def positional_encoding_table(d_model: int, max_len: int) -> np.ndarray:
    """fp32 restatement of `PositionalEncoding.__init__` (`main/model/mdm.py:377-384`).

    A real checkpoint stores this buffer; the synthetic state dict carries our own copy and
    both the reference (through load_state_dict) and this framework consume THAT copy."""
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2).astype(np.float32)
                      * np.float32(-np.log(10000.0) / d_model)).astype(np.float32)
    arg = (position * div_term).astype(np.float32)
    pe[:, 0::2] = np.sin(arg.astype(np.float64)).astype(np.float32)
    pe[:, 1::2] = np.cos(arg.astype(np.float64)).astype(np.float32)
    return pe[:, None, :].copy()          # [max_len, 1, d_model]


def rotary_inv_freq(dim: int) -> np.ndarray:
    """`SinusoidalEmbeddings.__init__` (`main/model/local_attention/rotary.py:11`), fp32."""
    e = (np.arange(0, dim, 2).astype(np.float32) / np.float32(dim)).astype(np.float32)
    return (np.float32(1.0) / np.power(np.float32(10000.0), e).astype(np.float32)).astype(np.float32)


def synth_state_dict(cfg: DSGConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    D, J, S, A, As = cfg.latent_dim, cfg.njoints, cfg.n_seed, cfg.audio_dim, cfg.audio_src_dim
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(prefix, n_out, n_in, scale=0.5):
        sd[prefix + ".weight"] = _mat(prefix + ".weight", (n_out, n_in), seed, scale)
        sd[prefix + ".bias"] = _vec(prefix + ".bias", n_out, seed)

    lin("WavEncoder.audio_feature_map", A, As)
    pe = positional_encoding_table(D, cfg.pe_max_len)
    sd["sequence_pos_encoder.pe"] = pe
    lin("input_process.poseEmbedding", D, J)
    for i in range(cfg.num_layers):
        p = f"seqTransEncoder.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = _mat(p + "self_attn.in_proj_weight", (3 * D, D), seed, 1.0)
        sd[p + "self_attn.in_proj_bias"] = _vec(p + "self_attn.in_proj_bias", 3 * D, seed)
        lin(p + "self_attn.out_proj", D, D, 1.0)
        lin(p + "linear1", cfg.ff_size, D, 1.0)
        lin(p + "linear2", D, cfg.ff_size, 1.0)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = _vec(p + n + ".weight", D, seed, offset=1.0)
            sd[p + n + ".bias"] = _vec(p + n + ".bias", D, seed)
    sd["embed_timestep.sequence_pos_encoder.pe"] = pe     # aliased buffer in the reference
    lin("embed_timestep.time_embed.0", D, D, 1.0)
    lin("embed_timestep.time_embed.2", D, D, 1.0)
    if cfg.variant == VARIANT_DSG:
        lin("embed_style", 64, cfg.style_dim_in, 1.0)
        lin("embed_text", D - 64, J * S, 1.0)
    else:
        lin("embed_style", D, cfg.style_dim_in, 1.0)
        lin("embed_text", A, J, 1.0)
        if cfg.variant == 5:
            lin("embed_text_last", A, J, 1.0)
    lin("output_process.poseFinal", J, D, 1.0)
    sd["rel_pos.inv_freq"] = rotary_inv_freq(D // cfg.local_heads)
    lin("input_process2", D, 2 * D + A, 1.0)
    return sd


def synth_window_inputs(cfg: DSGConfig, batch: int, window: int = 0, clip0: int = 0, seed_pose_scale=0.0, clips=None):
    """Synthetic per-window conditioning: WavLM-like features, style one-hot, seed poses.  Batch element b is clip
    `clips[b]` (default: the contiguous range clip0 + b) -- everything is a function of the clip id alone."""
    ids = [clip0 + b for b in range(batch)] if clips is None else [int(c) for c in clips]
    assert len(ids) == batch
    audio = np.stack([
        _feat(cfg, 1000 + 16 * c + window) for c in ids]).astype(np.float32)
    style = np.zeros((batch, cfg.style_dim_in), dtype=np.float32)
    style[:, 0] = 1.0
    if seed_pose_scale == 0.0:
        seedp = np.zeros((batch, cfg.njoints, 1, cfg.n_seed), dtype=np.float32)
    else:
        seedp = np.stack([
            (seed_pose_scale * np.random.RandomState(7 + c).randn(cfg.njoints, 1, cfg.n_seed))
            for c in ids]).astype(np.float32)
    mask_local = np.ones((1, cfg.n_poses), dtype=bool)
    y = {"audio": audio, "style": style, "seed": seedp, "mask_local": mask_local}
    if cfg.variant == 5:        # DSG++: the fixed "closing" pose snippet (BEAT-TWH sample.py:85-93)
        y["seed_last"] = np.stack([(0.2 * np.random.RandomState(70 + c).randn(cfg.njoints, 1, cfg.n_seed))
                                   for c in ids]).astype(np.float32)
    return y


def _feat(cfg: DSGConfig, s: int) -> np.ndarray:
    return np.random.RandomState(s).randn(cfg.audio_frames, cfg.audio_src_dim).astype(np.float32)


def synth_wavlm_state_dict(cfg: dict, seed: int = 0):
    """Seeded synthetic WavLM checkpoint of ANY size (no trained WavLM-Large.pt is available offline, and 315 M parameters do not
    fit a fixture): every tensor of `wavlm.wavlm_state_shapes(cfg)` from its own stream, like `synth_state_dict`.  Scales keep a
    24-layer pre-norm encoder well conditioned: matrices N(0, (0.5 / sqrt(fan_in))^2), LayerNorm scales 1 + 0.05 N, biases 0.02 N,
    the weight-norm gain of the positional convolution ~ 1, relative-position table 0.1 N, grep_a 1 + 0.05 N."""
    from .wavlm import wavlm_state_shapes
    sd = OrderedDict()
    for name, shape in wavlm_state_shapes(cfg).items():
        rs = _rs("wavlm." + name, seed)
        n = int(np.prod(shape))
        if name.endswith("weight_g"):
            v = 1.0 + 0.05 * rs.randn(*shape)
        elif name.endswith("grep_a"):
            v = 1.0 + 0.05 * rs.randn(*shape)
        elif "layer_norm" in name or name.endswith(".2.1.weight") or name.endswith(".2.1.bias") or name.endswith(".2.weight") or name.endswith(".2.bias"):
            v = (1.0 if name.endswith("weight") else 0.0) + 0.05 * rs.randn(*shape)
        elif name.endswith("bias"):
            v = 0.02 * rs.randn(*shape)
        elif name.endswith("relative_attention_bias.weight"):
            v = 0.1 * rs.randn(*shape)
        elif name == "mask_emb":
            v = rs.rand(*shape)
        else:
            fan_in = n // shape[0]
            # big matrices: float32 draws (half the time and memory of randn's float64)
            v = rs.standard_normal(size=shape).astype(np.float32) * np.float32((1.0 if "conv_layers" in name else 0.5) / np.sqrt(fan_in))
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd
