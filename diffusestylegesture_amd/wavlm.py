"""WavLM audio feature extraction for the sampling path's conditioning -- PyTorch-ROCm, inference only.

north_star keeps this stage on PyTorch-ROCm and asks for it to be *cached once per clip*: the reference runs the WavLM
encoder inside its window loop (`main/mydiffusion_zeggs/sample.py:251`, `wav2wavlm` `:44-48`) although the features do
not depend on the generated motion.  Here all windows of a clip go through ONE batched forward before the sampler starts
(`clip_features`), and the result is the `[K, n_poses, C]` array the sampler consumes (`generate_clip`).

This is an own implementation of the forward pass the reference reaches through `WavLM.extract_features(wav)` with its
default arguments (`WavLM/WavLM.py:318-376`), written against the checkpoint format (`{'cfg': {...}, 'model': state_dict}`,
loaded at `sample.py:30-41`) rather than against the module tree:

* conv feature extractor, 7 x (Conv1d -> [LayerNorm over channels | GroupNorm on block 0 only] -> GELU)
  (`WavLM.py:378-500`); keys `feature_extractor.conv_layers.{i}.0.weight`, norm at `.2.1.*` (layer_norm mode) / `.2.*`;
* LayerNorm over the conv channels, `post_extract_proj` (`WavLM.py:333-340`);
* convolutional positional embedding: grouped Conv1d with weight norm over the kernel axis (`weight_g`, `weight_v`),
  drop the last frame for an even kernel, GELU, added to the input (`WavLM.py:507-526`, `:565-567`);
* encoder layers, pre-norm (`layer_norm_first`, WavLM-Large) or post-norm (`WavLM.py:690-741`);
* self-attention with T5-style bucketed relative position bias owned by layer 0 and shared by all layers, gated per
  layer and per query by `grep_linear` / `grep_a` (`modules_WavLM.py:417-455`, `:509-530`).

Differences by design: q/k/v projections are one fused GEMM; the bias table is cached per sequence length; attention is
`scaled_dot_product_attention` with the gated bias as an additive mask (one fused kernel on ROCm); weight norm is folded
at load time; optional bf16 autocast for the encoder GEMMs (LayerNorm / softmax stay fp32).  Like the reference's
`wav2wavlm`, the waveform is NOT normalised before the extractor even when `cfg['normalize']` is set (`sample.py:44-48`
passes the raw 16 kHz samples).

Parity: `tests/test_wavlm.py` against `extract_features` of the imported reference for a WavLM-Large-like and a
WavLM-Base-like small configuration (fixture `tests/golden/g9_wavlm_small.npz`, generator `make_goldens.py wavlm`).
"""
from __future__ import annotations

import ast
import math

import numpy as np
import torch
import torch.nn.functional as F

_DEFAULTS = dict(  # WavLMConfig defaults (WavLM.py:160-212); a checkpoint's 'cfg' overrides them
    extractor_mode="default", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False,
    conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_bias=False, normalize=False,
    conv_pos=128, conv_pos_groups=16, relative_position_embedding=False, num_buckets=320, max_distance=1280,
    gru_rel_pos=False)


def _conv_spec(text):
    """'[(512,10,5)] + [(512,3,2)] * 4 + ...' -> list of (dim, kernel, stride) without eval()."""
    node = ast.parse(text, mode="eval").body

    def ev(n):
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Add):
            return ev(n.left) + ev(n.right)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Mult):
            l, r = n.left, n.right
            if isinstance(r, ast.Constant):
                return ev(l) * int(r.value)
            return int(l.value) * ev(r)
        if isinstance(n, ast.List):
            return [tuple(int(e.value) for e in t.elts) for t in n.elts]
        raise ValueError("unsupported conv_feature_layers expression")
    return ev(node)


def relative_position_buckets(n_q, n_k, num_buckets, max_distance):
    """Bidirectional T5-style buckets of (key position - query position) (modules_WavLM.py:417-442)."""
    rel = torch.arange(n_k)[None, :] - torch.arange(n_q)[:, None]
    half = num_buckets // 2
    bucket = (rel > 0).long() * half
    dist = rel.abs()
    exact = half // 2
    large = exact + (torch.log(dist.float() / exact) / math.log(max_distance / exact) * (half - exact)).long()
    large = torch.minimum(large, torch.full_like(large, half - 1))
    return bucket + torch.where(dist < exact, dist, large)


WAVLM_LARGE = dict(  # the topology of WavLM-Large.pt (the checkpoint sample.py:33 loads): 24 x 1024, 16 heads, ffn 4096, 315.5 M parameters
    extractor_mode="layer_norm", encoder_layers=24, encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16,
    layer_norm_first=True, normalize=True, conv_bias=False, conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2",
    conv_pos=128, conv_pos_groups=16, relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True)


def wavlm_state_shapes(cfg):
    """name -> shape of every tensor of the reference model's state dict for `cfg` (WavLM.py:214-316, modules_WavLM.py:300-415): the
    checkpoint contract this loader consumes, restated so that a synthetic checkpoint of any size can be made without the reference
    (diffusestylegesture_amd.synth.synth_wavlm_state_dict); make_goldens.py loads it into the reference model with strict=True."""
    c = dict(_DEFAULTS)
    c.update(cfg or {})
    C_, Fd, H = int(c["encoder_embed_dim"]), int(c["encoder_ffn_embed_dim"]), int(c["encoder_attention_heads"])
    sh = {"mask_emb": (C_,)}
    prev = 1
    ln_mode = c["extractor_mode"] == "layer_norm"
    for i, (dim, k, _) in enumerate(_conv_spec(c["conv_feature_layers"])):
        pre = f"feature_extractor.conv_layers.{i}."
        sh[pre + "0.weight"] = (dim, prev, k)
        if c["conv_bias"]:
            sh[pre + "0.bias"] = (dim,)
        if ln_mode:
            sh[pre + "2.1.weight"] = sh[pre + "2.1.bias"] = (dim,)
        elif i == 0:
            sh[pre + "2.weight"] = sh[pre + "2.bias"] = (dim,)
        prev = dim
    sh["layer_norm.weight"] = sh["layer_norm.bias"] = (prev,)
    if prev != C_:
        sh["post_extract_proj.weight"], sh["post_extract_proj.bias"] = (C_, prev), (C_,)
    K, G = int(c["conv_pos"]), int(c["conv_pos_groups"])
    sh["encoder.pos_conv.0.bias"], sh["encoder.pos_conv.0.weight_g"], sh["encoder.pos_conv.0.weight_v"] = (C_,), (1, 1, K), (C_, C_ // G, K)
    for l in range(int(c["encoder_layers"])):
        p = f"encoder.layers.{l}."
        a = p + "self_attn."
        if c["relative_position_embedding"] and l == 0:
            sh[a + "relative_attention_bias.weight"] = (int(c["num_buckets"]), H)
        if c["gru_rel_pos"]:
            sh[a + "grep_a"] = (1, H, 1, 1)
            sh[a + "grep_linear.weight"], sh[a + "grep_linear.bias"] = (8, C_ // H), (8,)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sh[a + n + ".weight"], sh[a + n + ".bias"] = (C_, C_), (C_,)
        sh[p + "self_attn_layer_norm.weight"] = sh[p + "self_attn_layer_norm.bias"] = (C_,)
        sh[p + "fc1.weight"], sh[p + "fc1.bias"] = (Fd, C_), (Fd,)
        sh[p + "fc2.weight"], sh[p + "fc2.bias"] = (C_, Fd), (C_,)
        sh[p + "final_layer_norm.weight"] = sh[p + "final_layer_norm.bias"] = (C_,)
    sh["encoder.layer_norm.weight"] = sh["encoder.layer_norm.bias"] = (C_,)
    return sh


class WavLMFeatures:
    """Inference-only WavLM encoder.  `extract_features(wav)` mirrors the call the reference makes on its model object
    (returns `(features [B, T', C], None)`), so `wav2wavlm(model, wav)` reads the same on both sides."""

    def __init__(self, cfg, state_dict, device="cpu", compute_dtype=torch.float32):
        c = dict(_DEFAULTS)
        c.update(cfg or {})
        if c["activation_fn"] != "gelu":
            raise NotImplementedError("only the gelu encoder of the released WavLM checkpoints is built")
        self.cfg = c
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(self.device, torch.float32)
              for k, v in state_dict.items()}
        self.conv = []
        ln_mode = c["extractor_mode"] == "layer_norm"
        for i, (dim, k, stride) in enumerate(_conv_spec(c["conv_feature_layers"])):
            pre = f"feature_extractor.conv_layers.{i}."
            blk = {"w": sd[pre + "0.weight"], "b": sd.get(pre + "0.bias"), "stride": stride, "norm": None}
            if ln_mode:
                blk["norm"] = ("layer", sd[pre + "2.1.weight"], sd[pre + "2.1.bias"])
            elif i == 0:
                blk["norm"] = ("group", sd[pre + "2.weight"], sd[pre + "2.bias"])
            self.conv.append(blk)
        self.feat_ln = (sd["layer_norm.weight"], sd["layer_norm.bias"])
        self.proj = (sd["post_extract_proj.weight"], sd["post_extract_proj.bias"]) if "post_extract_proj.weight" in sd else None
        # weight norm over the kernel axis (dim=2): w = v * g / ||v||, the norm taken over (out, in) per kernel tap
        v, g = sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"]
        self.pos_w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.pos_b = sd["encoder.pos_conv.0.bias"]
        self.pos_k, self.pos_groups = int(c["conv_pos"]), int(c["conv_pos_groups"])
        self.enc_ln = (sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"])
        self.H = int(c["encoder_attention_heads"])
        self.layers = []
        for l in range(int(c["encoder_layers"])):
            p = f"encoder.layers.{l}."
            a = p + "self_attn."
            lay = {
                "wqkv": torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]]),
                "bqkv": torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]]),
                "wo": sd[a + "out_proj.weight"], "bo": sd[a + "out_proj.bias"],
                "ln1": (sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"]),
                "ln2": (sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"]),
                "w1": sd[p + "fc1.weight"], "b1": sd[p + "fc1.bias"], "w2": sd[p + "fc2.weight"], "b2": sd[p + "fc2.bias"],
            }
            if c["gru_rel_pos"]:
                lay["gw"], lay["gb"], lay["ga"] = sd[a + "grep_linear.weight"], sd[a + "grep_linear.bias"], sd[a + "grep_a"]
            self.layers.append(lay)
        self.rel_table = sd.get("encoder.layers.0.self_attn.relative_attention_bias.weight") if c["relative_position_embedding"] else None
        self._bias_cache = {}

    @classmethod
    def from_checkpoint(cls, path, device="cpu", compute_dtype=torch.float32):
        """`WavLM-Large.pt` as loaded at sample.py:34-39: a dict with 'cfg' and 'model'."""
        ck = torch.load(path, map_location="cpu")
        return cls(ck["cfg"], ck["model"], device=device, compute_dtype=compute_dtype)

    # ------------------------------------------------------------------------------------------------------
    def _position_bias(self, n):
        if self.rel_table is None:
            return None
        if n not in self._bias_cache:
            b = relative_position_buckets(n, n, int(self.cfg["num_buckets"]), int(self.cfg["max_distance"])).to(self.device)
            self._bias_cache[n] = self.rel_table[b].permute(2, 0, 1).contiguous()          # [H, n, n]
        return self._bias_cache[n]

    def _frontend(self, wav):
        x = wav.unsqueeze(1)
        for blk in self.conv:
            x = F.conv1d(x, blk["w"], blk["b"], stride=blk["stride"])
            if blk["norm"] is not None:
                kind, g, b = blk["norm"]
                if kind == "layer":
                    x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), g, b).transpose(1, 2)
                else:
                    x = F.group_norm(x, x.shape[1], g, b)
            x = F.gelu(x)
        x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), *self.feat_ln)
        if self.proj is not None:
            x = F.linear(x, *self.proj)
        return x                                                                              # [B, T', C]

    def _attention(self, x, lay, bias):
        B, L, Cd = x.shape
        H, hd = self.H, Cd // self.H
        dt = self.compute_dtype
        qkv = F.linear(x.to(dt), lay["wqkv"].to(dt), lay["bqkv"].to(dt)).view(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
        mask = None
        if bias is not None:
            mask = bias.unsqueeze(0)
            if "gw" in lay:     # gate from the attention INPUT, per head and query (modules_WavLM.py:516-527)
                gate = torch.sigmoid(F.linear(x.view(B, L, H, hd).transpose(1, 2), lay["gw"], lay["gb"]).view(B, H, L, 2, 4).sum(-1))
                ga, gb = gate[..., :1], gate[..., 1:]
                mask = (ga * (gb * lay["ga"] - 1.0) + 2.0) * mask                             # [B, H, L, L]
            mask = mask.to(dt)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=mask)
        o = o.transpose(1, 2).reshape(B, L, Cd)
        return F.linear(o, lay["wo"].to(dt), lay["bo"].to(dt)).float()

    def _ffn(self, x, lay):
        dt = self.compute_dtype
        h = F.gelu(F.linear(x.to(dt), lay["w1"].to(dt), lay["b1"].to(dt)).float())
        return F.linear(h.to(dt), lay["w2"].to(dt), lay["b2"].to(dt)).float()

    @torch.no_grad()
    def extract_features(self, source, padding_mask=None, mask=False, ret_conv=False, output_layer=None, ret_layer_results=False):
        if padding_mask is not None or mask or output_layer is not None or ret_layer_results:
            raise NotImplementedError("only the plain inference call of sample.py:46 is built")
        wav = torch.as_tensor(source, dtype=torch.float32, device=self.device)
        if wav.dim() == 1:
            wav = wav[None]
        x = self._frontend(wav)
        if ret_conv:
            return x, None
        Cd = x.shape[-1]
        pc = F.conv1d(x.transpose(1, 2), self.pos_w, self.pos_b, padding=self.pos_k // 2, groups=self.pos_groups)
        if self.pos_k % 2 == 0:
            pc = pc[:, :, :-1]
        x = x + F.gelu(pc).transpose(1, 2)
        pre = bool(self.cfg["layer_norm_first"])
        if not pre:
            x = F.layer_norm(x, (Cd,), *self.enc_ln)
        bias = self._position_bias(x.shape[1])
        for lay in self.layers:
            if pre:
                x = x + self._attention(F.layer_norm(x, (Cd,), *lay["ln1"]), lay, bias)
                x = x + self._ffn(F.layer_norm(x, (Cd,), *lay["ln2"]), lay)
            else:
                x = F.layer_norm(x + self._attention(x, lay, bias), (Cd,), *lay["ln1"])
                x = F.layer_norm(x + self._ffn(x, lay), (Cd,), *lay["ln2"])
        if pre:
            x = F.layer_norm(x, (Cd,), *self.enc_ln)
        return x, None

    @torch.no_grad()
    def clip_features(self, windows, n_poses=88):
        """All windows of a clip in one batched forward -> [K, n_poses, C] (the per-clip cache).  Each window is what the
        reference hands to wav2wavlm one at a time (sample.py:233-251); the interpolation is sample.py:47."""
        wav = torch.stack([torch.as_tensor(np.asarray(w), dtype=torch.float32).reshape(-1) for w in windows]).to(self.device)
        rep = self.extract_features(wav)[0]
        return F.interpolate(rep.transpose(1, 2), size=n_poses, align_corners=True, mode="linear").transpose(1, 2).contiguous()


def wavlm_init(path="./WavLM/WavLM-Large.pt", device="cuda:0", compute_dtype=torch.float32):
    """Counterpart of sample.py:30-41."""
    return WavLMFeatures.from_checkpoint(path, device=device, compute_dtype=compute_dtype)


def wav2wavlm(model, wav_input_16khz, device=None, n_poses=88):
    """Counterpart of sample.py:44-48: [1, n] waveform -> [1, n_poses, C] features."""
    rep = model.extract_features(wav_input_16khz)[0]
    return F.interpolate(rep.transpose(1, 2), size=n_poses, align_corners=True, mode="linear").transpose(1, 2)
