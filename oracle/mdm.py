"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the MDM denoiser forward.

Restates, op for op, the live branches of the reference denoiser:
  * ZEGGS  `cross_local_attention3_style1` + `trans_enc`   `main/model/mdm.py:166-233`, `:357`
  * DSG+   `cross_local_attention4`                         `BEAT-TWH-main/model/mdm.py:134-146`, `:187-224`
  * TimestepEmbedder `main/model/mdm.py:434-448`, InputProcess `:461-467`, OutputProcess `:490-504`,
    WavEncoder `:545-552`
  * rotary `main/model/local_attention/rotary.py:8-27`
  * LocalAttention.forward `main/model/local_attention/local_attention.py:91-199`
  * nn.TransformerEncoderLayer (post-norm, erf-GELU, eps 1e-5) / nn.MultiheadAttention -- third-party
    arithmetic (PyTorch, pinned torch==1.9 in `requirements.txt:1`, not under /root/reference):
    restated from its documented semantics and pinned by goldens generated with the torch in this
    image (tests/golden/make_goldens.py).
Pinned against the imported reference by tests/golden/g2_*.npz, g5_*.npz (see tests/test_oracle_vs_golden.py).
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package; the
product path (diffusestylegesture_amd/) never does.
"""
from __future__ import annotations

import math

import numpy as np

try:  # vectorised erf for GELU: torch's CPU kernel is ~10x faster than scipy's and equally exact; values only
    import torch as _torch

    def _erf(a):
        return _torch.erf(_torch.from_numpy(np.ascontiguousarray(a))).numpy()
except Exception:  # pragma: no cover
    try:
        from scipy.special import erf as _erf
    except Exception:
        _erf = np.vectorize(math.erf)

FLT_MAX = float(np.finfo(np.float32).max)


def _lin(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def _layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * g + b


def _gelu(x):
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype)


def _softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def _rotary(x, inv_freq):
    """x: [..., n, hd]; positions 0..n-1 (rotary.py:12-27).  freqs are formed in fp32 like the reference."""
    n, hd = x.shape[-2], x.shape[-1]
    t = np.arange(n, dtype=np.float32)
    fr = (t[:, None] * inv_freq.astype(np.float32)[None, :]).astype(np.float32)
    fr = np.concatenate([fr, fr], -1)                      # [n, hd]
    if x.dtype == np.float32:
        c, s = np.cos(fr), np.sin(fr)                      # fp32 like torch
    else:
        c, s = np.cos(fr.astype(x.dtype)), np.sin(fr.astype(x.dtype))
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    rot = np.concatenate([-x2, x1], -1)
    return x * c + rot * s


def local_attention(q, window, mask=None):
    """LocalAttention.forward with q=k=v, causal, look_backward=1, look_forward=0, exact_windowsize=False.

    q: [b, n, d] (b = batch*heads); mask: bool [mb, n] with b % mb == 0 (or None)."""
    b, n, d = q.shape
    dt = q.dtype
    w = n // window
    assert w * window == n
    scale = dt.type(d ** -0.5)
    bq = q.reshape(b, w, window, d)
    pad = np.full((b, 1, window, d), -1, dtype=dt)                      # look_around pad_value=-1
    padded = np.concatenate([pad, bq], 1)
    bk = np.concatenate([padded[:, 0:w], padded[:, 1:w + 1]], 2)        # [b, w, 2*window, d]
    bv = bk
    seq = np.arange(n).reshape(1, w, window)
    pseq = np.concatenate([np.full((1, 1, window), -1), seq], 1)
    bq_k = np.concatenate([pseq[:, 0:w], pseq[:, 1:w + 1]], 2)          # [1, w, 2*window]
    sim = np.einsum("bhie,bhje->bhij", bq, bk) * scale
    mask_value = dt.type(-np.finfo(dt).max)
    causal = seq[..., :, None] < bq_k[..., None, :]
    sim = np.where(causal, mask_value, sim)
    if mask is not None:
        mb = mask.shape[0]
        assert b % mb == 0
        h = b // mb
        m = mask.reshape(mb, w, window)
        mp = np.concatenate([np.zeros((mb, 1, window), bool), m], 1)
        mk = np.concatenate([mp[:, 0:w], mp[:, 1:w + 1]], 2)[:, :, None, :]   # [mb, w, 1, 2*window]
        mk = np.repeat(mk, h, axis=0)                                    # 'b ... -> (b h) ...'
        sim = np.where(~mk, mask_value, sim)
    attn = _softmax(sim)
    out = np.einsum("bhij,bhje->bhie", attn, bv)
    return out.reshape(b, n, d)


class MDMOracle:
    def __init__(self, state_dict, cfg, dtype=np.float32):
        self.cfg = cfg
        self.dt = np.dtype(dtype)
        self.sd = {k: np.asarray(v).astype(self.dt) for k, v in state_dict.items()
                   if k != "rel_pos.inv_freq"}
        self.inv_freq = np.asarray(state_dict["rel_pos.inv_freq"]).astype(np.float32)
        self.probes = {}

    # -- pieces -------------------------------------------------------------------------------
    def timestep_embed(self, timesteps):
        sd = self.sd
        pe = sd["sequence_pos_encoder.pe"][np.asarray(timesteps), 0, :]           # [B, D]
        h = _lin(pe, sd["embed_timestep.time_embed.0.weight"], sd["embed_timestep.time_embed.0.bias"])
        h = h / (1.0 + np.exp(-h))                                                # SiLU
        return _lin(h, sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"])

    def _encoder_layer(self, x, i):
        sd, cfg = self.sd, self.cfg
        p = f"seqTransEncoder.layers.{i}."
        B, n, D = x.shape
        H = cfg.num_heads
        hd = D // H
        qkv = _lin(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        sh = lambda t: t.reshape(B, n, H, hd).transpose(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        att = _softmax((q @ k.transpose(0, 1, 3, 2)) * self.dt.type(1.0 / math.sqrt(hd)))
        o = (att @ v).transpose(0, 2, 1, 3).reshape(B, n, D)
        o = _lin(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = _layer_norm(x + o, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        f = _lin(_gelu(_lin(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        return _layer_norm(x + f, sd[p + "norm2.weight"], sd[p + "norm2.bias"])

    # -- forward ------------------------------------------------------------------------------
    def forward(self, x, timesteps, y, uncond_info=False):
        """x [B,J,1,T]; timesteps [B] int; y: dict(style [B,sd], seed [B,J,1,S], audio [B,Ta,A_src],
        mask_local bool [1|B, T]) -> [B,J,1,T]"""
        cfg, sd, dt = self.cfg, self.sd, self.dt
        x = np.asarray(x).astype(dt)
        B, J, _, T = x.shape
        D, Hl = cfg.latent_dim, cfg.local_heads
        emb_t = self.timestep_embed(timesteps)                                    # [B, D]
        style = np.asarray(y["style"]).astype(dt)
        seed = np.asarray(y["seed"]).astype(dt)
        audio = np.asarray(y["audio"]).astype(dt)
        if uncond_info:
            style_e = np.zeros((B, cfg.tok_style_dim), dt)
        else:
            style_e = _lin(style, sd["embed_style.weight"], sd["embed_style.bias"])
        if cfg.variant == 3:
            seed_in = np.zeros((B, J * cfg.n_seed), dt) if uncond_info else seed[:, :, 0, :].reshape(B, -1)
            text = _lin(seed_in, sd["embed_text.weight"], sd["embed_text.bias"])
            emb_1 = np.concatenate([style_e, text], 1)                            # [B, D]
            enc = _lin(audio, sd["WavEncoder.audio_feature_map.weight"],
                       sd["WavEncoder.audio_feature_map.bias"])                   # [B, T, A]
        else:
            emb_1 = style_e                                                       # [B, D]
            text = _lin(seed[:, :, 0, :].transpose(0, 2, 1), sd["embed_text.weight"],
                        sd["embed_text.bias"])                                    # [B, S, A]
            enc_a = _lin(audio, sd["WavEncoder.audio_feature_map.weight"],
                         sd["WavEncoder.audio_feature_map.bias"])                 # [B, T-S, A]
            parts = [text, enc_a]
            if cfg.variant == 5:                                                  # BEAT-TWH mdm.py:227-230
                last = np.asarray(y["seed_last"]).astype(dt)
                parts.append(_lin(last[:, :, 0, :].transpose(0, 2, 1), sd["embed_text_last.weight"],
                                  sd["embed_text_last.bias"]))                    # [B, S, A]
            enc = np.concatenate(parts, 1)                                        # [B, T, A]
        tok = emb_1 + emb_t                                                       # [B, D]
        xf = x[:, :, 0, :].transpose(0, 2, 1)                                     # [B, T, J]
        x_ = _lin(xf, sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"])
        cat = np.concatenate([np.repeat(tok[:, None, :], T, 1), x_, enc], -1)     # [B, T, 2D+A]
        h = _lin(cat, sd["input_process2.weight"], sd["input_process2.bias"])     # [B, T, D]
        self.probes["after_input_process2"] = h
        hd = D // Hl
        hh = h.reshape(B, T, Hl, hd).transpose(0, 2, 1, 3).reshape(B * Hl, T, hd)
        hh = _rotary(hh, self.inv_freq).astype(dt)
        mask = y.get("mask_local", None)
        mask = None if mask is None else np.asarray(mask).astype(bool)
        hh = local_attention(hh, cfg.window, mask)
        h = hh.reshape(B, Hl, T, hd).transpose(0, 2, 1, 3).reshape(B, T, D)
        self.probes["after_local_attention"] = h
        xs = np.concatenate([tok[:, None, :], h], 1)                              # [B, T+1, D]
        xh = xs.reshape(B, T + 1, Hl, hd).transpose(0, 2, 1, 3).reshape(B * Hl, T + 1, hd)
        xh = _rotary(xh, self.inv_freq).astype(dt)
        xs = xh.reshape(B, Hl, T + 1, hd).transpose(0, 2, 1, 3).reshape(B, T + 1, D)
        self.probes["encoder_in"] = xs
        for i in range(cfg.num_layers):
            xs = self._encoder_layer(xs, i)
            self.probes[f"after_layer{i}"] = xs
        out = _lin(xs[:, 1:], sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"])
        return np.ascontiguousarray(out.transpose(0, 2, 1))[:, :, None, :].astype(dt)

    __call__ = forward
