#!/usr/bin/env python3
"""Headline benchmark: gesture frames/sec, 1000-step DDPM, 320-frame ZEGGS clip (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input = sampling ONE 320-frame clip per GPU
(4 windows x 1000 denoising steps, batch 1 = BASELINE config[1]).  Clips are independent, so N GPUs run N clips
(weak scaling, no collective on the data path); the finished poses are gathered to rank 0 with one RCCL gather inside
the timed region.  Inputs (synthetic WavLM features, synthetic weights) are resident in HBM when the clock starts.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # see diffusestylegesture_amd/__init__.py

import numpy as np
import torch


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3, help="clips per GPU inside the timed region")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--sampler", default="ddpm", choices=["ddpm", "ddim50"])
    p.add_argument("--batch", type=int, default=1, help="clips advanced in lock step per GPU")
    p.add_argument("--steps-per-graph", type=int, default=0)
    p.add_argument("--config", default="zeggs", choices=["zeggs", "beat", "twh"],
                   help="zeggs = headline (BASELINE config[1]); beat/twh = DiffuseStyleGesture+ dims, 1830-frame clip (config[4])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-steps", type=int, default=400)
    return p.parse_args()


def cpu_baseline(n_steps):
    """Times the CPU oracle (numpy restatement of the reference path, validated against goldens) on a bounded sample:
    n_steps DDPM steps of one ZEGGS window, batch 1; extrapolated to 4 x 1000 steps per 320-frame clip."""
    from diffusestylegesture_amd import config as C
    from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
    from oracle import sampler
    from oracle.mdm import MDMOracle
    from oracle.schedule import OracleDiffusion
    cfg = C.ZEGGS
    m = MDMOracle(synth_state_dict(cfg, 20240), cfg)
    y = synth_window_inputs(cfg, 1, window=0)
    shape = (1, cfg.njoints, 1, cfg.n_poses)
    d = OracleDiffusion()
    nf = sampler.philox_noise_fn(shape, 1, 0)
    sampler.p_sample_loop(d, m, shape, nf, {"y": y}, skip_timesteps=995)         # warm-up
    # BLAS thread count: the GEMMs of one step are small (89 x 256 x 1024 at most), so all host threads is not the
    # fastest setting; a short sweep picks the best one and `cores` reports the threads actually used
    cores, limiter = os.cpu_count() or 1, None
    try:
        from threadpoolctl import threadpool_limits
        best = None
        for nt in sorted({1, 4, 8, 16, 32, cores}):
            if nt > cores:
                continue
            with threadpool_limits(limits=nt):
                t0 = time.perf_counter()
                sampler.p_sample_loop(d, m, shape, nf, {"y": y}, skip_timesteps=980)
                dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        cores = best[1]
        limiter = threadpool_limits(limits=cores)
    except Exception:
        pass
    t0 = time.perf_counter()
    sampler.p_sample_loop(d, m, shape, nf, {"y": y}, skip_timesteps=1000 - n_steps)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    ms_step = 1000.0 * dt / n_steps
    return {"value": round(320.0 / (4000 * ms_step / 1000.0), 3), "unit": "frames/s", "cores": int(cores),
            "kind": "port", "ms_per_denoise_step": round(ms_step, 3),
            "sample": f"{n_steps} DDPM steps of one 88-frame ZEGGS window (batch 1, fp32 numpy oracle), "
                      f"extrapolated to 4x1000 steps per 320-frame clip"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")          # RCCL on ROCm
    from diffusestylegesture_amd import config as C
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import DSGDenoiser
    from diffusestylegesture_amd.sample import generate_clip
    from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

    cfg = C.CONFIGS[a.config]
    B = a.batch
    model = DSGDenoiser(cfg, precision=a.precision, max_batch=B, device=local, steps_per_graph=a.steps_per_graph)
    model.load_state_dict(synth_state_dict(cfg, 20240))
    diffusion = create_gaussian_diffusion("ddim50" if a.sampler == "ddim50" else "")
    sample_fn = diffusion.ddim_sample_loop if a.sampler == "ddim50" else diffusion.p_sample_loop
    if a.config == "zeggs":
        n_windows = 4
        frames_per_clip = n_windows * cfg.stride                                # 320 nominal (312 emitted)
    else:
        frames_per_clip = 1830                                                  # BEAT-TWH sample.py:56, max_len=0
        n_windows = -(-frames_per_clip // cfg.stride)                           # ceil -> 16 windows
    # synthetic per-window audio features, resident in HBM before the clock starts (clip index = rank*B + b)
    feats = [torch.from_numpy(synth_window_inputs(cfg, B, window=w, clip0=rank * B)["audio"]).cuda(local)
             for w in range(n_windows)]
    style = [1] + [0] * (cfg.style_dim_in - 1)

    def one_clip(i):
        if a.config == "zeggs":
            return generate_clip(model, diffusion, feats, style, seed=123456 + i, smoothing=True, sample_fn=sample_fn,
                                 stream_id=rank)
        from diffusestylegesture_amd.sample import generate_clip_dsgplus
        seed0 = torch.from_numpy(synth_window_inputs(cfg, B, window=0, clip0=rank * B, seed_pose_scale=0.1)["seed"]).cuda(local)
        return generate_clip_dsgplus(model, diffusion, feats, style, seed0, frames_per_clip, seed=123456 + i,
                                     sample_fn=sample_fn, stream_id=rank)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        one_clip(i)
    sync()
    step_us = []
    t0 = time.perf_counter()
    poses = None
    for i in range(a.steps):
        poses = one_clip(a.warmup + i)
        step_us.append(diffusion.last_step_time_us())
    if dist is not None:        # the only exchange of the path: finished poses -> rank 0 (RCCL over xGMI)
        from diffusestylegesture_amd.parallel import gather_poses
        gather_poses(poses, world * B, dist, dst=0, device=f"cuda:{local}")
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        total_frames = world * B * a.steps * frames_per_clip
        value = total_frames / dt
        n_denoise = diffusion.num_timesteps
        us = float(np.mean(step_us))
        # algorithmic bytes per denoising step (SURVEY s8d / DESIGN.md): per-step weights in the compute dtype +
        # fp32 state I/O (x_t in, noise in, x_{t-1} out) per clip in the batch
        # per-step weight parameters / fp32 state bytes per clip
        wparams, sbytes = {"zeggs": (7.183e6, 1.205e6), "beat": (13.25e6, 3.694e6), "twh": (20.23e6, 4.018e6)}[a.config]
        wbytes = wparams * (2 if a.precision == "bf16" else 4)
        abytes = wbytes + sbytes * B
        achieved = abytes / (us * 1e-6) / 1e9
        # HBM-side traffic per step from the rocprofv3 PMC passes (tools/pmc_traffic.py; FETCH_SIZE doubled as the
        # MI355X guide prescribes for wide coalesced reads), measured for the headline configuration only
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_zeggs_b1_bf16.json")))   # newest round last
        tf = cands[-1] if cands else ""
        if a.config == "zeggs" and a.precision == "bf16" and B == 1 and os.path.exists(tf):
            traffic = json.load(open(tf))["traffic_bytes_per_step_fetch_x2"]
        out = {
            "metric": (f"gesture frames/sec, {'1000-step DDPM' if a.sampler == 'ddpm' else '50-step DDIM'}, "
                       + ("320-frame ZEGGS clip" if a.config == "zeggs" else f"1830-frame {a.config.upper()} clip (DSG+)")),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1000.0 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"1xMI355X per rank, batch={B}, {frames_per_clip}-frame {a.config.upper()} clip "
                                   f"({n_windows} windows x {n_denoise} denoising steps), {a.sampler.upper()} {a.precision}",
                       "clips_per_gpu": B, "frames_emitted_per_clip": int(poses.shape[1]),
                       "denoise_steps_per_window": n_denoise, "parallelism": f"clips x{world}"},
            "us_per_denoise_step": round(us, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "algorithmic_bytes_per_denoise_step": abytes,
                         "note": "one denoising step = 2 + 3*L dependent kernel dispatches (batch-1 latency mode), submitted as "
                                 "hand-written AQL packets on the library's own HSA queue (DSG_AQL=0: HIP launches); achieved = "
                                 "algorithmic bytes / time per step, timed from the first doorbell to the completion signal "
                                 "of the last packet (HIP events around the loop on the HIP-launch path)"},
        }
        if world == 1 and not a.no_cpu_baseline and a.config == "zeggs" and a.sampler == "ddpm":
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
