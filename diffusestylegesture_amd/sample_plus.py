"""DiffuseStyleGesture+ / ++ command line (BEAT / TWH): counterpart of `BEAT-TWH-main/mydiffusion_beat_twh/sample.py`.

The reference's `main()` (`:194-268`) builds the per-frame conditioning from a wav + transcript (WavLM, word vectors,
onsets, ... `process_*_bvh.py`, dataset pipelines that are not part of this path) or reads it pre-extracted from an h5
file (`--tst_path`), then calls `inference()` (`:44-192`).  This module takes the same flags, expects the pre-extracted
per-frame features as an array (`--features_npy`, `[n_frames, audio_feature_dim]`: what the reference calls `textaudio`),
the seed-gesture snippet (`--seed_npy`, `[n_seed + 2, motion_dim]` raw poses: what it loads from
`<dataset>_dataset/processed/gesture_<dataset>/<clip>.npy`, `:112-129`) and runs `inference()`'s window loop on the HIP
path (`sample.generate_clip_dsgplus`), de-normalises with the dataset statistics (`process/gesture_*_{mean,std}_v0.npy`,
kept as data in `data/beat_twh_mean_std.npz`) and writes the `[n_frames, motion_dim]` poses; the dataset-specific BVH
pipelines (`pose2bvh_bugfix` with sklearn pipelines `.sav`) stay in the reference."""
from __future__ import annotations

import argparse
import math
import os

import numpy as np

from .config import DSGPLUS_CONFIGS
from .sample import generate_clip_dsgplus


def seed_features(seed_gesture, mean, std):
    """poses + velocities + accelerations of the normalised seed snippet (sample.py:125-129): [n_seed+2, m] -> [1, 3m, 1, n_seed]"""
    g = (np.asarray(seed_gesture, np.float64) - mean) / std
    vel = g[1:] - g[:-1]
    acc = vel[1:] - vel[:-1]
    f = np.concatenate((g[2:], vel[1:], acc), axis=1).astype(np.float32)
    return np.ascontiguousarray(f.T[None, :, None, :])


def window_features(textaudio, n_frames, stride):
    """sample.py:52-73: ceil(n / stride) windows, zero-padded tail.  Returns ([K, stride, C], real_n_frames)."""
    ta = np.asarray(textaudio, np.float32)
    if n_frames:
        ta = ta[:n_frames]
    real_n = ta.shape[0]
    k = 1 if real_n < stride else math.ceil(real_n / stride)
    pad = np.zeros((k * stride - real_n, ta.shape[1]), np.float32)
    return np.concatenate((ta, pad), 0).reshape(k, stride, ta.shape[1]), real_n


def build_parser():
    p = argparse.ArgumentParser(description='DiffuseStyleGesture')          # flags of BEAT-TWH sample.py:275-289
    p.add_argument('--config', default='./configs/DiffuseStyleGesture.yml')
    p.add_argument('--gpu', type=str, default='0')
    p.add_argument('--tst_prefix', nargs='+')
    p.add_argument('--no_cuda', type=list, default=['0'])
    p.add_argument('--model_path', type=str, default='./model000450000.pt')
    p.add_argument('--tst_path', type=str, default=None)
    p.add_argument('--wav_path', type=str, default=None)
    p.add_argument('--txt_path', type=str, default=None)
    p.add_argument('--save_dir', type=str, default='sample_dir')
    p.add_argument('--max_len', type=int, default=0)
    p.add_argument('--skip_timesteps', type=int, default=0)
    p.add_argument('--dataset', type=str, default='BEAT')
    p.add_argument('--wavlm_path', type=str, default='./WavLM/WavLM-Large.pt')
    p.add_argument('--word2vector_path', type=str, default='./crawl-300d-2M.vec')
    # framework additions
    p.add_argument('--name', default='DiffuseStyleGesture+', choices=['DiffuseStyleGesture', 'DiffuseStyleGesture+', 'DiffuseStyleGesture++'],
                   help="`name` of the reference's DiffuseStyleGesture.yml: attention3 / attention4 / attention5 (sample.py:297-303)")
    p.add_argument('--features_npy', required=True, help='[n_frames, audio_feature_dim] per-frame conditioning (the reference\'s textaudio)')
    p.add_argument('--seed_npy', required=True, help='[n_seed + 2, motion_dim] raw seed poses (sample.py:112-124)')
    p.add_argument('--seed_last_npy', default='', help='DiffuseStyleGesture++: raw poses of the closing snippet (sample.py:85-93)')
    p.add_argument('--speaker', type=int, default=0, help='index of the one-hot style / speaker entry')
    p.add_argument('--precision', default='bf16', choices=['bf16', 'bf16w2', 'fp32'],
                   help='bf16 (default); bf16w2 = bf16 activations, weights as hi + lo bf16 (3x closer to fp32); fp32 = the reference arithmetic')
    p.add_argument('--version', default='v0', choices=['v0', 'v2'],
                   help="`version` of the reference's yml (sample.py:309-315): v0 = poses + velocities + accelerations (njoints = 3 x motion_dim), "
                        "v2 (BEAT only) = njoints = motion_dim = 1141")
    p.add_argument('--mean_std_npz', default='', help="v2: file with `mean` and `std` [motion_dim] (the reference's gesture_BEAT_mean_v2.npy / _std_v2.npy)")
    return p


def main(argv=None):
    import torch
    from .diffusion import create_gaussian_diffusion
    from .model import DSGDenoiser
    args = build_parser().parse_args(argv)
    if args.wav_path or args.txt_path or args.tst_path:
        raise SystemExit("feature extraction from wav / transcript / h5 stays in the reference's pipelines: pass --features_npy")
    if args.dataset not in ('BEAT', 'TWH') or (args.name, args.dataset, args.version) not in DSGPLUS_CONFIGS:
        raise NotImplementedError(f"{args.dataset} {args.version}")          # sample.py:323-327 (TWH has no v2 branch either)
    cfg = DSGPLUS_CONFIGS[(args.name, args.dataset, args.version)]           # every name x dataset pair of sample.py:297-323
    if args.version == 'v2':
        # the v2 statistics are not part of the reference tree; and its seed construction (pose + velocity + acceleration =
        # 3 x motion_dim columns, sample.py:125-129) cannot produce njoints = motion_dim features: the seed file holds the
        # [n_seed, njoints] feature rows themselves
        if not args.mean_std_npz:
            raise SystemExit("--version v2 needs --mean_std_npz (gesture_BEAT_mean_v2 / _std_v2 of the dataset)")
        ms = np.load(args.mean_std_npz)
        mean, std = ms["mean"], ms["std"]
    else:
        ms = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "beat_twh_mean_std.npz"))
        mean, std = ms[args.dataset + "_mean"], ms[args.dataset + "_std"]
    v2_seed = lambda a: np.ascontiguousarray(((np.asarray(a, np.float64)[: cfg.n_seed] - mean) / std).astype(np.float32).T[None, :, None, :])
    dev = int(args.gpu)
    torch.cuda.set_device(dev)
    model = DSGDenoiser(cfg, precision=args.precision, max_batch=1, device=dev)
    model.load_state_dict(torch.load(args.model_path, map_location='cpu'))
    diffusion = create_gaussian_diffusion()
    wins, real_n = window_features(np.load(args.features_npy), args.max_len, cfg.stride)
    feats = [torch.from_numpy(w[None]).cuda(dev) for w in wins]
    mk_seed = v2_seed if args.version == 'v2' else (lambda a: seed_features(a[: cfg.n_seed + 2], mean, std))
    seed0 = torch.from_numpy(mk_seed(np.load(args.seed_npy))).cuda(dev)
    seed_last = None
    if cfg.variant == 5:
        seed_last = torch.from_numpy(mk_seed(np.load(args.seed_last_npy))).cuda(dev)
    style = np.zeros(cfg.style_dim_in, np.float32)
    style[args.speaker] = 1.0
    seq = generate_clip_dsgplus(model, diffusion, feats, style, seed0, real_n, seed=123456, skip_timesteps=args.skip_timesteps,
                                seed_last=seed_last, feature_division=1 if args.version == 'v2' else 3)[0]
    out_poses = np.multiply(seq, std) + mean                                  # sample.py:184 (no clipping of std here)
    os.makedirs(args.save_dir, exist_ok=True)
    stem = os.path.join(args.save_dir, os.path.splitext(os.path.basename(args.features_npy))[0])
    np.save(stem + "_poses.npy", out_poses)
    print(stem + "_poses.npy", out_poses.shape)
    return stem + "_poses.npy"


if __name__ == '__main__':
    main()
