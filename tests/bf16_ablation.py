#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a study, not a collected test): where does the bf16 drift of the headline configuration come from?

The HIP bf16 path rounds to bf16 at a fixed list of points (operands of the MFMA contractions; everything else -- residual stream,
LayerNorm statistics, local attention, conditioning, the sampler update -- is fp32).  This script restates the fp32 numpy oracle
with a switch per rounding point (round-to-nearest-even to 8 mantissa bits, products / sums in fp32 like the MFMA), runs a whole
1000-step DDPM window of the headline workload (ZEGGS, batch 1, the Philox noise of the goldens) per variant and reports the
relative L2 distance to the pure fp32 oracle:

    python tests/bf16_ablation.py [--steps 1000] [--windows 1] [--only] [--except]

  all          every point on  (= what the HIP bf16 path does; cross-checked on the GPU: tests/test_gpu_round4.py)
  only:<p>     only point p on
  except:<p>   every point but p

Points: weights (all packed matrices), state (x_t as the pose embedding's operand), x0a (encoder input into the layer-0 QKV), ln2
(LayerNorm2 rows into QKV / the pose head), qk (Q and K as stored), v (V as stored), p (softmax numerators into the PV product),
attn (attention rows into out_proj), ln1 (LayerNorm1 rows into linear1), hidden (GELU output into linear2).
Result of the run behind DESIGN.md s2 (round 4): profiles/r04_bf16_ablation_oracle.log."""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusestylegesture_amd import config as C                       # noqa: E402
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs      # noqa: E402
from oracle import mdm as M                                          # noqa: E402
from oracle import sampler                                           # noqa: E402
from oracle.schedule import OracleDiffusion                          # noqa: E402

POINTS = ["weights", "state", "x0a", "ln2", "qk", "v", "p", "attn", "ln1", "hidden"]
# "weights" by matrix (only:w_in ... ; `weights` = all six): pose embedding (folded), in_proj, out_proj, linear1, linear2, pose head
WPOINTS = {"w_in": (), "w_qkv": ("in_proj_weight",), "w_o": ("out_proj.weight",), "w_1": ("linear1.weight",), "w_2": ("linear2.weight",),
           "w_out": ("poseFinal.weight",)}


def bf16(x):
    """fp32 -> bf16 -> fp32, round to nearest even (v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


class RoundedOracle(M.MDMOracle):
    """MDMOracle (ZEGGS branch) with bf16 rounding at the points in `on`; the arithmetic between them is the oracle's."""

    def __init__(self, sd, cfg, on):
        super().__init__(sd, cfg)
        self.on = set(on)
        s = self.sd
        D = cfg.latent_dim
        W2 = s["input_process2.weight"].astype(np.float64)
        # the library folds input_process2[:, D:2D] . poseEmbedding in fp64 and packs THAT (dsg_hip.cpp: finalize_weights)
        self.Wfold = (W2[:, D:2 * D] @ s["input_process.poseEmbedding.weight"].astype(np.float64)).astype(np.float32)
        self.cbase = (W2[:, D:2 * D] @ s["input_process.poseEmbedding.bias"].astype(np.float64) + s["input_process2.bias"]).astype(np.float32)
        self.W2a, self.W2c = s["input_process2.weight"][:, :D], s["input_process2.weight"][:, 2 * D:]
        if "weights" in self.on or "w_in" in self.on:
            self.Wfold = bf16(self.Wfold)
        tags = [t for w, ts in WPOINTS.items() if w in self.on or "weights" in self.on for t in ts]
        for k in list(s):
            if any(t in k for t in tags):
                s[k] = bf16(s[k])

    def R(self, name, x):
        return bf16(x) if name in self.on else x

    def _encoder_layer(self, x, i, first):
        sd, cfg = self.sd, self.cfg
        p = f"seqTransEncoder.layers.{i}."
        B, n, D = x.shape
        H = cfg.num_heads
        hd = D // H
        qkv = M._lin(self.R("x0a" if first else "ln2", x), sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = self.R("qk", qkv[..., :D]), self.R("qk", qkv[..., D:2 * D]), self.R("v", qkv[..., 2 * D:])
        sh = lambda t: t.reshape(B, n, H, hd).transpose(0, 2, 1, 3)
        q, k, v = sh(q), sh(k), sh(v)
        s = (q @ k.transpose(0, 1, 3, 2)) * np.float32(1.0 / math.sqrt(hd))
        e = np.exp(s - s.max(-1, keepdims=True))
        o = (self.R("p", e) @ v) / e.sum(-1, keepdims=True)          # the kernels normalise after the PV product, sum from fp32 numerators
        o = o.transpose(0, 2, 1, 3).reshape(B, n, D)
        o = M._lin(self.R("attn", o), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = M._layer_norm(x + o, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        hid = M._gelu(M._lin(self.R("ln1", x), sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
        f = M._lin(self.R("hidden", hid), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        return M._layer_norm(x + f, sd[p + "norm2.weight"], sd[p + "norm2.bias"])

    def forward(self, x, timesteps, y, uncond_info=False):
        cfg, sd, dt = self.cfg, self.sd, self.dt
        assert cfg.variant == 3 and not uncond_info
        x = np.asarray(x).astype(dt)
        B, J, _, T = x.shape
        D, Hl = cfg.latent_dim, cfg.local_heads
        emb_t = self.timestep_embed(timesteps)
        style_e = M._lin(np.asarray(y["style"]).astype(dt), sd["embed_style.weight"], sd["embed_style.bias"])
        text = M._lin(np.asarray(y["seed"]).astype(dt)[:, :, 0, :].reshape(B, -1), sd["embed_text.weight"], sd["embed_text.bias"])
        tok = np.concatenate([style_e, text], 1) + emb_t
        enc = M._lin(np.asarray(y["audio"]).astype(dt), sd["WavEncoder.audio_feature_map.weight"], sd["WavEncoder.audio_feature_map.bias"])
        xf = x[:, :, 0, :].transpose(0, 2, 1)
        h = self.R("state", xf) @ self.Wfold.T + (tok @ self.W2a.T)[:, None, :] + enc @ self.W2c.T + self.cbase
        hd = D // Hl
        hh = h.reshape(B, T, Hl, hd).transpose(0, 2, 1, 3).reshape(B * Hl, T, hd)
        hh = M._rotary(hh, self.inv_freq).astype(dt)
        mask = y.get("mask_local", None)
        hh = M.local_attention(hh, cfg.window, None if mask is None else np.asarray(mask).astype(bool))
        h = hh.reshape(B, Hl, T, hd).transpose(0, 2, 1, 3).reshape(B, T, D)
        xs = np.concatenate([tok[:, None, :], h], 1)
        xh = xs.reshape(B, T + 1, Hl, hd).transpose(0, 2, 1, 3).reshape(B * Hl, T + 1, hd)
        xs = M._rotary(xh, self.inv_freq).astype(dt).reshape(B, Hl, T + 1, hd).transpose(0, 2, 1, 3).reshape(B, T + 1, D)
        for i in range(cfg.num_layers):
            xs = self._encoder_layer(xs, i, i == 0)
        out = M._lin(self.R("ln2", xs[:, 1:]), sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"])
        return np.ascontiguousarray(out.transpose(0, 2, 1))[:, :, None, :].astype(dt)

    __call__ = forward


def run(on, sd, cfg, steps, windows, seed):
    m = RoundedOracle(sd, cfg, on)
    d = OracleDiffusion()
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    outs, prev = [], None
    for w in range(windows):
        y = synth_window_inputs(cfg, 1, window=w)
        if prev is not None:
            y["seed"] = prev[..., -cfg.n_seed:]
        base = w * (1 + steps)                                    # one Philox stream runs through the windows of a clip
        nf = (lambda b: (lambda dr: sampler.philox.normal_bj1t(shape, seed, b + dr, 0)))(base)
        prev = sampler.p_sample_loop(d, m, shape, nf, {"y": y}, skip_timesteps=1000 - steps)
        outs.append(prev)
    return np.concatenate(outs, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--windows", type=int, default=1)
    ap.add_argument("--seed", type=int, default=123456)
    ap.add_argument("--modes", default="all,only,except")
    a = ap.parse_args()
    cfg = C.ZEGGS
    sd = synth_state_dict(cfg, 20240)
    from threadpoolctl import threadpool_limits
    rel = lambda x, r: float(np.linalg.norm(x.astype(np.float64) - r) / np.linalg.norm(r))
    with threadpool_limits(limits=8):
        t0 = time.time()
        ref = run([], sd, cfg, a.steps, a.windows, a.seed).astype(np.float64)
        print(f"fp32 oracle: {a.windows} window(s) x {a.steps} steps in {time.time() - t0:.1f} s", flush=True)
        variants = []
        if "all" in a.modes:
            variants.append(("all", POINTS))
        if "only" in a.modes:
            variants += [("only:" + p, [p]) for p in POINTS]
        if "wsplit" in a.modes:
            variants += [("only:" + p, [p]) for p in WPOINTS] + [("all-but-weights", [q for q in POINTS if q != "weights"]),
                                                                 ("all, w_out+w_in fp32", [q for q in POINTS if q != "weights"] + ["w_qkv", "w_o", "w_1", "w_2"]),
                                                                 ("all, w_1+w_2 fp32", [q for q in POINTS if q != "weights"] + ["w_qkv", "w_o", "w_in", "w_out"])]
        if "except" in a.modes:
            variants += [("except:" + p, [q for q in POINTS if q != p]) for p in POINTS]
        for name, on in variants:
            out = run(on, sd, cfg, a.steps, a.windows, a.seed)
            print(f"{name:16s} rel-L2 vs fp32 oracle = {rel(out, ref):.3e}", flush=True)


if __name__ == "__main__":
    main()
