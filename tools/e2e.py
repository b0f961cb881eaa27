#!/usr/bin/env python3
"""End to end on one MI355X, the stages of the reference's `main()` for one clip (main/mydiffusion_zeggs/sample.py:299-395):

    .wav (16 kHz, 16 s = a 320-frame clip) -> window split (sample.py:214-249)
         -> WavLM-Large features, ALL windows in one batched forward = the per-clip cache   [PyTorch-ROCm, north_star keeps it there]
         -> 4 windows x 1000 DDPM steps through libdsg_hip.so                                [the hot path: bench.py's metric]
         -> de-normalise + Savitzky-Golay + .bvh text                                        [C++ behind the C ABI]

No trained checkpoints exist offline: the WavLM weights are the seeded synthetic checkpoint of the REAL WavLM-Large topology
(24 x 1024, 315.5 M parameters; tests/golden/g16_wavlm_large.npz pins that forward to the imported reference), the denoiser's are
`synth_state_dict(ZEGGS)`; the audio is seeded noise written to a real .wav file and read back.  Model loading (checkpoint -> GPU) is
reported but not part of the per-clip time, as in the reference's main().  Prints one JSON line.

    python tools/e2e.py [--reps 3] [--wavlm-dtype fp32|bf16] [--precision bf16|fp32]"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.bvh import pose2bvh
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.sample import denormalise, generate_clip, load_wav_16k, window_audio
from diffusestylegesture_amd.synth import synth_state_dict, synth_wavlm_state_dict
from diffusestylegesture_amd.wavlm import WAVLM_LARGE, WavLMFeatures

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--wavlm-dtype", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
a = ap.parse_args()
cfg = C.ZEGGS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ms = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "zeggs_mean_std.npz"))

t0 = time.perf_counter()
wsd = synth_wavlm_state_dict(WAVLM_LARGE, 5)
t_synth = time.perf_counter() - t0
t0 = time.perf_counter()
wavlm = WavLMFeatures(WAVLM_LARGE, wsd, device="cuda:0", compute_dtype=torch.bfloat16 if a.wavlm_dtype == "bf16" else torch.float32)
del wsd
model = DSGDenoiser(cfg, precision=a.precision, max_batch=1, device=0)
model.load_state_dict(synth_state_dict(cfg, 20240))
diffusion = create_gaussian_diffusion()
torch.cuda.synchronize()
t_load = time.perf_counter() - t0

with tempfile.TemporaryDirectory() as td:
    from scipy.io import wavfile
    wav_path = os.path.join(td, "015_Happy_4_x_1_0.wav")
    wavfile.write(wav_path, 16000, (np.random.RandomState(3).randn(16 * 16000) * 0.1 * 32767).astype(np.int16))
    stages = []
    for rep in range(a.reps + 1):          # the first pass warms up (MIOpen / hipBLASLt heuristics of the conv extractor, queues, code objects)
        t = [time.perf_counter()]
        wav = load_wav_16k(wav_path)
        wins, n_frames = window_audio(wav, 0, cfg.n_poses, cfg.n_seed)
        t.append(time.perf_counter())
        feats = wavlm.clip_features(wins, cfg.n_poses)                       # [K, 88, 1024] on the GPU: the per-clip cache
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        poses = generate_clip(model, diffusion, [f[None] for f in feats], [1, 0, 0, 0, 0, 0], seed=123456, smoothing=True)[0]
        t.append(time.perf_counter())
        out = os.path.join(td, "out.bvh")
        pose2bvh(denormalise(poses, ms["mean"], ms["std"]), out, length=poses.shape[0], smoothing=True)
        t.append(time.perf_counter())
        if rep:
            stages.append(np.diff(t))
    bvh_bytes = os.path.getsize(out)
st = np.array(stages).min(0) * 1e3
total = float(st.sum())
print(json.dumps({
    "what": "wav -> windows -> WavLM-Large-topology features (one batched forward) -> 4 x 1000 DDPM steps -> .bvh, one 320-frame ZEGGS clip, 1 x MI355X",
    "windows": len(wins), "frames_nominal": 320, "frames_emitted": int(poses.shape[0]), "denoiser_precision": a.precision, "wavlm_gemm_dtype": a.wavlm_dtype,
    "ms": {"read_wav_and_window": round(float(st[0]), 2), "wavlm_features_all_windows": round(float(st[1]), 2), "sampling_4x1000_steps": round(float(st[2]), 2),
           "denormalise_and_bvh": round(float(st[3]), 2), "total": round(total, 2)},
    "frames_per_s_end_to_end": round(320.0 / (total * 1e-3), 1), "frames_per_s_sampling_only": round(320.0 / (float(st[2]) * 1e-3), 1),
    "us_per_denoise_step": round(diffusion.last_step_time_us(), 2), "sample_path": diffusion.last_sample_path(),
    "model_load_s": {"synthesise_wavlm_checkpoint_on_host": round(t_synth, 1), "weights_to_gpu_and_pack": round(t_load, 2)},
    "bvh_bytes": bvh_bytes, "reps": a.reps}), flush=True)
