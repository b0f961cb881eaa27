"""Clip orchestration and CLI of the sampling path -- the caller contract of the reference.

Mirrors `inference()` / `main()` / the argparse block of `main/mydiffusion_zeggs/sample.py:210-420` (ZEGGS) and
`BEAT-TWH-main/mydiffusion_beat_twh/sample.py:44-192` (DSG+): window split, per-window conditioning, seed hand-off,
root-position continuity, the one-frame blend (the reference's `len(last_poses) == 1` quirk), stitching and
de-normalisation.  The denoising itself is `sample_fn(model, shape, ...)` = `DSGDiffusion.p_sample_loop`, i.e. the
HIP library.  Windows of one clip are serially dependent (window c is seeded by window c-1), clips are independent.
"""
from __future__ import annotations

import argparse
import math
import os

import numpy as np

from . import lib as L

style2onehot = {
    'Happy': [1, 0, 0, 0, 0, 0], 'Sad': [0, 1, 0, 0, 0, 0], 'Neutral': [0, 0, 1, 0, 0, 0],
    'Old': [0, 0, 0, 1, 0, 0], 'Angry': [0, 0, 0, 0, 1, 0], 'Relaxed': [0, 0, 0, 0, 0, 1],
}


def _xp(use_torch):
    if use_torch:
        import torch
        return torch
    return None


def _zeggs_window_y(cfg, feat, sty, prev, seed_pose, use_torch, mask):
    """model_kwargs['y'] of one ZEGGS window (sample.py:227-251): seed poses = zeros / the caller's for window 0, the previous
    window's last n_seed frames (post-stitch) afterwards."""
    S, J = cfg.n_seed, cfg.njoints
    B = int(feat.shape[0])
    if prev is not None:
        seedp = prev[..., -S:].contiguous() if use_torch else np.ascontiguousarray(prev[..., -S:])
    elif seed_pose is not None:
        seedp = seed_pose
    elif use_torch:
        import torch
        seedp = torch.zeros(B, J, 1, S, device=feat.device)
    else:
        seedp = np.zeros((B, J, 1, S), np.float32)
    return {"style": sty, "seed": seedp, "audio": feat, "mask_local": mask}


def _zeggs_stitch(out, s, S, smoothing, use_torch):
    """sample.py:269-289: cut the overlap off the previous window, root-position continuity, the one-frame blend."""
    if out:
        last = out[-1][..., -S:]
        last = last.clone() if use_torch else last.copy()
        out[-1] = out[-1][..., :-S]
        if smoothing:
            delta = (s[:, 0:3, :, 0] - last[:, 0:3, :, 0])[..., None]
            s[:, 0:3] = s[:, 0:3] - delta
        # `for j in range(len(last_poses))` with len() == batch dim of a [1, J, 1, S] tensor: only frame 0
        s[..., 0] = last[..., 0] * 0.5 + s[..., 0] * 0.5
    out.append(s)


def _zeggs_finish(out, S, use_torch):
    out[-1] = out[-1][..., :-S]
    if use_torch:
        import torch
        seq = torch.cat([o[:, :, 0, :] for o in out], dim=2).permute(0, 2, 1)      # [B, K*stride, J]
        seq = seq[:, S:].contiguous()
        if seq.is_cuda:          # device -> PINNED host memory (torch's caching host allocator recycles the block): 23 MB per 16 clips, 1.5 -> 0.9 ms
            host = torch.empty(seq.shape, dtype=seq.dtype, pin_memory=True)
            host.copy_(seq, non_blocking=True)
            torch.cuda.current_stream(seq.device).synchronize()
            seq = host.numpy()
        else:
            seq = seq.numpy()
    else:
        seq = np.concatenate([o[:, :, 0, :] for o in out], axis=2).transpose(0, 2, 1)[:, S:]
    return np.ascontiguousarray(seq, dtype=np.float32)


def _style_batch(style, B, use_torch, dev=None):
    if L.is_torch(style):        # a tensor the caller keeps on the device: no host -> device copy per call (a small pageable copy costs
        sty = style.float()      # ~1.8 ms on ROCm -- 4 % of a 50-step DDIM pass of 4 windows, tools/prof_host.py)
        if sty.ndim == 1:
            sty = sty[None].expand(B, -1)
        return sty.to(dev).contiguous() if dev is not None else sty.contiguous()
    sty = np.asarray(style, np.float32)
    if sty.ndim == 1:
        sty = np.repeat(sty[None], B, 0)
    if use_torch:
        import torch
        return torch.from_numpy(sty).to(dev)
    return sty


def generate_clip(model, diffusion, feats, style, seed=123456, smoothing=True, skip_timesteps=0, sample_fn=None,
                  stream_id=0, seed_pose=None, device=None):
    """ZEGGS window loop (sample.py:236-296).  feats: sequence of K per-window WavLM features, each [B, T, A_src]
    (torch cuda tensors or numpy); style: one-hot list or [B, 6] array.  Returns normalised poses
    [B, K*stride - n_seed, J] (numpy float32) -- B independent clips advance in lock step."""
    cfg = model.cfg
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    use_torch = L.is_torch(feats[0])
    B = int(feats[0].shape[0])
    sample_fn = sample_fn or diffusion.p_sample_loop
    diffusion.manual_seed(seed, stream_id)          # torch.manual_seed(seed) at sample.py:212
    shape = (B, J, 1, T)
    out = []
    if use_torch:
        import torch
        mask = torch.ones(1, T, dtype=torch.bool, device=feats[0].device)
    else:
        mask = np.ones((1, T), bool)
    sty = _style_batch(style, B, use_torch, feats[0].device if use_torch else None)
    for c, feat in enumerate(feats):
        y = _zeggs_window_y(cfg, feat, sty, out[-1] if out else None, seed_pose, use_torch, mask)
        s = sample_fn(model, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=skip_timesteps,
                      init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        _zeggs_stitch(out, s, S, smoothing, use_torch)
    return _zeggs_finish(out, S, use_torch)


def generate_clips_streams(lanes, diffusion, feats_per_lane, styles, seed=123456, smoothing=True, skip_timesteps=0,
                           stream_ids=None, ddim=False, eta=0.0, kernel_set="recommended"):
    """Several clips of one GPU advanced concurrently on sampling LANES ("one clip per stream", BASELINE config[3]): `lanes`
    are N DSGDenoiser lanes over one copy of the weights (`model.clone()`); lane i samples the B clips of
    feats_per_lane[i] (K per-window features [B, T, A_src]; B = 1: one clip per lane) on its own HSA queue and the library
    interleaves the lanes' step loops (DSGDiffusion.p_sample_loop_multi).  Same window loop / stitching as `generate_clip`;
    lane i uses the Philox stream (seed, stream_ids[i]) and is bit-identical to `generate_clip(lanes[i], ..., stream_id=
    stream_ids[i])` run alone on the same lane.  `kernel_set`: "recommended" applies the set measured fastest for this many
    lanes x this batch to every lane (sticky: `DSGDenoiser.set_kernel_set`), None leaves the lanes as they are, a name forces
    that set.  The command processor serves one queue per compute pipe: up to 4 lanes overlap, more than 4 share pipes and
    block each other (measured: 4 lanes 2.7x one lane, 8 lanes slower than one) -- put the remaining clips into the lanes'
    batches.  Returns [N * B, K*stride - n_seed, J], lane-major."""
    n = len(lanes)
    cfg = lanes[0].cfg
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    K = len(feats_per_lane[0])
    if any(len(f) != K for f in feats_per_lane) or len(feats_per_lane) != n:
        raise ValueError("one feature list per lane, the same number of windows each")
    use_torch = L.is_torch(feats_per_lane[0][0])
    B = int(feats_per_lane[0][0].shape[0])
    stream_ids = list(range(n)) if stream_ids is None else list(stream_ids)
    with _lane_kernel_sets(lanes, B, kernel_set):
        diffusion.manual_seed(seed, 0)
        shape = (B, J, 1, T)
        dev = feats_per_lane[0][0].device if use_torch else None
        if use_torch:
            import torch
            mask = torch.ones(1, T, dtype=torch.bool, device=dev)
        else:
            mask = np.ones((1, T), bool)
        per_lane_style = (not L.is_torch(styles)) and np.asarray(styles).ndim == 2 and len(styles) == n and np.asarray(styles).shape[0] == n and B == 1
        stys = [_style_batch(styles[i] if per_lane_style else styles, B, use_torch, dev) for i in range(n)]
        outs = [[] for _ in range(n)]
        for c in range(K):
            ys = [{"y": _zeggs_window_y(cfg, feats_per_lane[i][c], stys[i], outs[i][-1] if outs[i] else None, None, use_torch, mask)}
                  for i in range(n)]
            ss = diffusion.p_sample_loop_multi(list(lanes), shape, ys, seeds=[seed] * n, stream_ids=stream_ids,
                                               skip_timesteps=skip_timesteps, ddim=ddim, eta=eta)
            for i in range(n):
                _zeggs_stitch(outs[i], ss[i], S, smoothing, use_torch)
    return np.concatenate([_zeggs_finish(o, S, use_torch) for o in outs], axis=0)


class _lane_kernel_sets:
    """The kernel set of a multi-lane call: "recommended" = the set measured fastest for this many lanes x this batch (it differs
    from what one lane alone would pick), a name forces that set, None leaves the lanes alone.  A set is a sticky
    property of a lane (lanes[0] is normally the caller's own model), so whatever the call changes is put back on exit -- the
    caller's later single-lane calls run the set they ran before (round-3 advisor)."""

    def __init__(self, lanes, batch, kernel_set):
        self.lanes, self.batch, self.want, self.saved = list(lanes), batch, kernel_set, None

    def __enter__(self):
        if self.want is not None:
            self.saved = [ln.kernel_set() for ln in self.lanes]
            ks = self.lanes[0].recommend_kernel_set(self.batch, len(self.lanes)) if self.want == "recommended" else self.want
            for ln in self.lanes:
                ln.set_kernel_set(ks)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            for ln, ks in zip(self.lanes, self.saved):
                ln.set_kernel_set(ks)
        return False


def _dsgplus_window_y(cfg, feats, c, sty, seedp, seed_last, use_torch, mask):
    """model_kwargs['y'] of window c of a DSG+ clip (BEAT-TWH sample.py:98-140) for the three model names of that tree."""
    S = cfg.n_seed
    feat = feats[c]
    seedp = seedp.contiguous() if use_torch else np.ascontiguousarray(seedp)
    y = {"style": sty, "seed": seedp, "audio": feat, "mask_local": mask}
    if cfg.variant == 3:
        # name "DiffuseStyleGesture" of the BEAT-TWH tree (attention3): S frames of left context in front of the window's
        # features -- zeros for window 0, the tail of the previous window's features afterwards (sample.py:100-102, :132-134)
        if use_torch:
            import torch
            left = torch.zeros_like(feat[:, :S]) if c == 0 else feats[c - 1][:, -S:]
            y["audio"] = torch.cat((left, feat), 1).contiguous()
        else:
            left = np.zeros_like(feat[:, :S]) if c == 0 else feats[c - 1][:, -S:]
            y["audio"] = np.ascontiguousarray(np.concatenate((left, feat), 1))
    if cfg.variant == 5:
        if seed_last is None:
            raise KeyError("seed_last")
        a = feat[:, :-S]
        y["audio"] = a.contiguous() if use_torch else np.ascontiguousarray(a)
        y["seed_last"] = seed_last
    return y


def _dsgplus_stitch(out, s, S, use_torch):
    """BEAT-TWH sample.py:150-160: cut the overlap off the previous window, the one-frame blend; no root shift."""
    if out:
        last = out[-1][..., -S:]
        last = last.clone() if use_torch else last.copy()
        out[-1] = out[-1][..., :-S]
        s[..., 0] = last[..., 0] * 0.5 + s[..., 0] * 0.5
    out.append(s)


def _dsgplus_finish(out, S, J, real_n_frames, feature_division, use_torch):
    if use_torch:
        import torch
        seq = torch.cat([o[:, :, 0, :] for o in out], dim=2).permute(0, 2, 1).contiguous().cpu().numpy()
    else:
        seq = np.concatenate([o[:, :, 0, :] for o in out], axis=2).transpose(0, 2, 1)
    seq = seq[:, S:][:, :real_n_frames]
    # "v0" data: the model features are poses + velocities + accelerations, only the poses are kept (motion_feature_division = 3,
    # BEAT-TWH sample.py:173-180); "v2": the whole vector (division 1)
    return np.ascontiguousarray(seq[:, :, : J // feature_division], dtype=np.float32)


def generate_clips_streams_dsgplus(lanes, diffusion, feats_per_lane, styles, seed0s, real_n_frames, seed=123456, skip_timesteps=0,
                                   stream_ids=None, seed_lasts=None, feature_division=3, ddim=False, eta=0.0,
                                   kernel_set="recommended"):
    """`generate_clips_streams` for the DSG+ window loop (BEAT-TWH sample.py:98-192; all three model names of that tree): lane i
    samples the B clips of feats_per_lane[i] (K per-window features), seeded by seed0s[i] [B, J, 1, S] (and seed_lasts[i] for
    DiffuseStyleGesture++), on its own HSA queue; the lanes' step loops are interleaved by the library.  Lane i is bit-identical
    to `generate_clip_dsgplus(lanes[i], ..., stream_id=stream_ids[i])` run alone on the same lane under the same kernel set.
    Returns [N * B, real_n_frames, J // feature_division], lane-major."""
    n = len(lanes)
    cfg = lanes[0].cfg
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    K = len(feats_per_lane[0])
    if any(len(f) != K for f in feats_per_lane) or len(feats_per_lane) != n or len(seed0s) != n:
        raise ValueError("one feature list and one seed clip per lane, the same number of windows each")
    use_torch = L.is_torch(feats_per_lane[0][0])
    B = int(feats_per_lane[0][0].shape[0])
    stream_ids = list(range(n)) if stream_ids is None else list(stream_ids)
    with _lane_kernel_sets(lanes, B, kernel_set):
        diffusion.manual_seed(seed, 0)
        shape = (B, J, 1, T)
        dev = feats_per_lane[0][0].device if use_torch else None
        if use_torch:
            import torch
            mask = torch.ones(1, T, dtype=torch.bool, device=dev)
        else:
            mask = np.ones((1, T), bool)
        sty = _style_batch(styles, B, use_torch, dev)
        outs = [[] for _ in range(n)]
        for c in range(K):
            ys = [{"y": _dsgplus_window_y(cfg, feats_per_lane[i], c, sty, seed0s[i] if c == 0 else outs[i][-1][..., -S:],
                                          None if seed_lasts is None else seed_lasts[i], use_torch, mask)} for i in range(n)]
            ss = diffusion.p_sample_loop_multi(list(lanes), shape, ys, seeds=[seed] * n, stream_ids=stream_ids,
                                               skip_timesteps=skip_timesteps, ddim=ddim, eta=eta)
            for i in range(n):
                _dsgplus_stitch(outs[i], ss[i], S, use_torch)
    return np.concatenate([_dsgplus_finish(o, S, J, real_n_frames, feature_division, use_torch) for o in outs], axis=0)


def generate_clip_dsgplus(model, diffusion, feats, style, seed0, real_n_frames, seed=123456, skip_timesteps=0,
                          sample_fn=None, stream_id=0, seed_last=None, feature_division=3):
    """DSG+ window loop (BEAT-TWH sample.py:98-192), attention4: zero-padded tail, no left audio context, GT seed for
    window 0, no root shift, last window kept whole, first S frames dropped, crop, keep the first J/3 features.
    `model.cfg.variant == 3` is that tree's "DiffuseStyleGesture" (attention3 at BEAT dims): S frames of left audio context.
    DiffuseStyleGesture++ (attention5, model.cfg.variant == 5): `feats` are still the stride-long windows; the last S
    feature frames of every window are dropped (sample.py:104, :138) and `seed_last` [B, J, 1, S] -- the same snippet
    for every window (sample.py:85-93) -- is passed as y['seed_last']."""
    cfg = model.cfg
    S, T, J = cfg.n_seed, cfg.n_poses, cfg.njoints
    use_torch = L.is_torch(feats[0])
    B = int(feats[0].shape[0])
    sample_fn = sample_fn or diffusion.p_sample_loop
    diffusion.manual_seed(seed, stream_id)
    shape = (B, J, 1, T)
    out = []
    sty = _style_batch(style, B, use_torch, feats[0].device if use_torch else None)
    if use_torch:
        import torch
        dev = feats[0].device
        mask = torch.ones(1, T, dtype=torch.bool, device=dev)
    else:
        mask = np.ones((1, T), bool)
    for c in range(len(feats)):
        y = _dsgplus_window_y(cfg, feats, c, sty, seed0 if c == 0 else out[-1][..., -S:], seed_last, use_torch, mask)
        s = sample_fn(model, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=skip_timesteps,
                      init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        _dsgplus_stitch(out, s, S, use_torch)
    return _dsgplus_finish(out, S, J, real_n_frames, feature_division, use_torch)


def window_audio(audio, n_frames, n_poses=88, n_seed=8, sr=16000, fps=20):
    """Audio slices per window with the n_seed-frame left context (sample.py:214-249): zeros for window 0, the
    previous chunk's tail otherwise.  Returns (list of float32 arrays of (n_poses * sr/fps) samples, n_frames)."""
    if n_frames == 0:
        n_frames = audio.shape[0] * fps // sr
    stride = n_poses - n_seed
    if n_frames < stride:
        k = 1
    else:
        k = math.floor(n_frames / stride)
        n_frames = k * stride
    spf = sr // fps
    audio = np.asarray(audio[: n_frames * spf], np.float32)
    chunks = audio.reshape(k, stride * spf)
    outs = []
    for c in range(k):
        left = np.zeros(n_seed * spf, np.float32) if c == 0 else chunks[c - 1][-n_seed * spf:]
        outs.append(np.concatenate([left, chunks[c]]))
    return outs, n_frames


def load_wav_16k(path):
    """Mono float32 waveform at 16 kHz in [-1, 1] (what `librosa.load(path, sr=16000)` returns, sample.py:346; librosa is
    not a dependency here: scipy reads the file and resamples polyphase when the file's rate differs)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    sr, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    else:
        x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != 16000:
        g = math.gcd(int(sr), 16000)
        x = resample_poly(x, 16000 // g, int(sr) // g).astype(np.float32)
    return x


def denormalise(poses, mean, std):
    """sample.py:320-326: std clipped at 0.01."""
    return np.multiply(poses, np.clip(std, a_min=0.01, a_max=None)) + mean


def inference(args, wavlm_model, audio, sample_fn, model, n_frames=0, smoothing=False, SG_filter=False,
              minibatch=False, skip_timesteps=0, n_seed=8, style=None, seed=123456, *, diffusion=None,
              wav2wavlm=None, mean=None, std=None, pose_writer=None, save_path=None):
    """Same positional signature as the reference `inference()` (sample.py:210).  `wav2wavlm(wavlm_model, wav)` must
    return the [1, n_poses, 1024] WavLM features of one window (the WavLM encoder stays on PyTorch-ROCm, outside this
    path); `pose_writer(out_poses, path, length, smoothing)` is the BVH writer.  Returns de-normalised poses."""
    if not minibatch:
        raise NotImplementedError("only the minibatch (windowed) path of inference() is on the sampling path")
    if diffusion is None:
        diffusion = sample_fn.__self__
    wins, n_frames = window_audio(audio, n_frames, args.n_poses, n_seed)
    feats = [wav2wavlm(wavlm_model, w) for w in wins]
    poses = generate_clip(model, diffusion, feats, style, seed=seed, smoothing=smoothing,
                          skip_timesteps=skip_timesteps, sample_fn=sample_fn)[0]
    out_poses = denormalise(poses, mean, std) if mean is not None else poses
    if pose_writer is not None and save_path is not None:
        pose_writer(out_poses, save_path, length=n_frames - n_seed, smoothing=SG_filter)
    return out_poses


def build_parser():
    p = argparse.ArgumentParser(description='DiffuseStyleGesture')          # flags of sample.py:400-407
    p.add_argument('--config', default='./configs/DiffuseStyleGesture.yml')
    p.add_argument('--gpu', type=str, default='0')
    p.add_argument('--no_cuda', type=list, default=['2'])
    p.add_argument('--model_path', type=str, default='./model000450000.pt')
    p.add_argument('--audiowavlm_path', type=str, default='')
    p.add_argument('--max_len', type=int, default=0)
    # framework additions
    p.add_argument('--precision', default='bf16', choices=['bf16', 'bf16w2', 'fp32'],
                   help='bf16 (default); bf16w2 = bf16 activations, weights as hi + lo bf16 (3x closer to fp32); fp32 = the reference arithmetic')
    p.add_argument('--features_npy', default='', help='pre-extracted WavLM features [K, n_poses, 1024] (the per-clip cache)')
    p.add_argument('--wavlm_path', default='./WavLM/WavLM-Large.pt', help='WavLM checkpoint (sample.py:33)')
    p.add_argument('--save_dir', default='sample_dir')
    p.add_argument('--timestep_respacing', default='')
    return p


def main(argv=None):
    import yaml
    import torch
    from .config import ZEGGS
    from .diffusion import create_gaussian_diffusion
    from .model import DSGDenoiser
    args = build_parser().parse_args(argv)
    cfg_yaml = {}
    if os.path.exists(args.config):
        with open(args.config) as f:
            cfg_yaml = yaml.safe_load(f) or {}
    n_poses = int(cfg_yaml.get("n_poses", ZEGGS.n_poses))
    assert n_poses == ZEGGS.n_poses
    dev = int(args.gpu)
    torch.cuda.set_device(dev)
    model = DSGDenoiser(ZEGGS, precision=args.precision, max_batch=1, device=dev)
    state_dict = torch.load(args.model_path, map_location='cpu')
    model.load_state_dict(state_dict)
    diffusion = create_gaussian_diffusion(args.timestep_respacing)
    name = os.path.basename(args.audiowavlm_path or args.features_npy)
    style = style2onehot[name.split('_')[1]]                              # sample.py:378
    os.makedirs(args.save_dir, exist_ok=True)
    if args.features_npy:
        feats = np.load(args.features_npy).astype(np.float32)
    else:
        # WavLM stage (PyTorch-ROCm, outside the HIP path): all windows of the clip in ONE batched forward, cached per clip
        from .wavlm import wavlm_init
        wav = load_wav_16k(args.audiowavlm_path)                          # librosa.load(path, sr=16000), sample.py:346
        wins, _ = window_audio(wav, args.max_len, n_poses, ZEGGS.n_seed)
        wavlm = wavlm_init(args.wavlm_path, device=f"cuda:{dev}")
        feats = wavlm.clip_features(wins, n_poses).cpu().numpy()
        np.save(os.path.join(args.save_dir, os.path.splitext(name)[0] + "_wavlm.npy"), feats)
        del wavlm
    if args.max_len:
        feats = feats[: max(1, args.max_len // (n_poses - ZEGGS.n_seed))]
    feats_t = [torch.from_numpy(f[None]).cuda(dev) for f in feats]
    poses = generate_clip(model, diffusion, feats_t, style, seed=123456, smoothing=True)[0]
    stem = os.path.join(args.save_dir, os.path.splitext(name)[0])
    np.save(stem + "_poses.npy", poses)
    # de-normalise (sample.py:320-326) and write the .bvh (process_zeggs_bvh.py:219) like the reference's main()
    from .bvh import pose2bvh
    ms = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "zeggs_mean_std.npz"))
    out_poses = denormalise(poses, ms["mean"], ms["std"])
    pose2bvh(out_poses, stem + ".bvh", length=out_poses.shape[0], smoothing=True)      # C++ writer (csrc/dsg_bvh.cpp)
    print(stem + ".bvh", out_poses.shape)
    return stem + ".bvh"


if __name__ == '__main__':
    main()
