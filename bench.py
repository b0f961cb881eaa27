#!/usr/bin/env python3
"""Headline benchmark: gesture frames/sec, 1000-step DDPM, 320-frame ZEGGS clip (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: re-executes itself under
                                                              torch.distributed.run with N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input = sampling `--clips-per-gpu` 320-frame clips per GPU
(4 windows x 1000 denoising steps each).  Default: 1 clip at batch 1 = BASELINE config[1].  `--clips-per-gpu 16` is the
per-GPU share of config[3] (8 GPUs x 16 = 128 clips); how the clips of a GPU are arranged:
    (default)         4 sampling lanes (own HSA queue each, ONE copy of the weights) x 4 clips per lane in lock step, the
                      lanes' step loops interleaved by the library (dsg_sample_multi) -- the fastest arrangement measured
    --lanes L         L lanes x clips-per-gpu / L clips each
    --mode streams    one clip per lane ("one clip per stream"): beyond 4 lanes the queues share hardware pipes and the
                      rate collapses -- kept for the measurement
    --mode lockstep   one batch of all clips advanced in lock step (the batched kernel set)
Clips are independent, so N GPUs run N x clips-per-gpu clips (weak scaling, no collective on the data path); the finished
poses are gathered to rank 0 with one RCCL gather inside the timed region.  Inputs (synthetic WavLM features, synthetic
weights) are resident in HBM when the clock starts.  De-normalisation + .bvh writing (C++, rank 0) is timed separately
(`postprocess_ms_per_clip`, `value_end_to_end`).  Prints ONE JSON line on rank 0.

The default line (`python bench.py`, 1 GPU, config[1]) also carries time-bounded SUB-RECORDS for the other BASELINE configs
(`--sub-records off` drops them; each has its own value / us_per_denoise_step / kernel_set / roofline, about 25 s in all):
    config2   50-step DDIM, batch 16 in lock step                      (BASELINE config[2])
    config3   16 clips as 4 lanes x batch 4 = config[3]'s per-GPU share (with --gpus N: N x 16 clips gathered over RCCL)
    config4   DiffuseStyleGesture+ BEAT and TWH denoisers, batch 1, 2 of the 16 windows of an 1830-frame clip (config[4]);
              `beat_64clips`: 64 BEAT clips per GPU as 4 lanes x batch 16, 1 of the 16 windows
    stream    256 clips per GPU as 4 lanes x batch 64 (the STREAM kernel set), 1 pass
    precision config[1] in the two other arithmetic modes: fp32 (the reference's own arithmetic) and bf16w2 (hi + lo bf16), 1 pass each
Top-level `value` / `config` stay config[1].  Sub-records with a committed PMC pass of their per-lane arrangement
(profiles/r*_traffic_<config>_b<B>_<set>_bf16.json, tools/measure_traffic.sh) carry it as `roofline.traffic`.
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # see diffusestylegesture_amd/__init__.py


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3, help="passes (clips per lane / batch element) inside the timed region")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--precision", default="bf16", choices=["bf16", "bf16w2", "fp32"])
    p.add_argument("--sampler", default="ddpm", choices=["ddpm", "ddim50"])
    p.add_argument("--clips-per-gpu", type=int, default=0, help="clips in flight per GPU (default 1; 16 = config[3]'s per-GPU share)")
    p.add_argument("--mode", default="auto", choices=["auto", "streams", "lockstep"],
                   help="several clips per GPU: one clip per lane / HSA queue, or one lock-step batch (auto: streams)")
    p.add_argument("--lanes", type=int, default=0, help="sampling lanes (HSA queues) per GPU; each lane advances clips-per-gpu / lanes "
                                                       "clips in lock step (streams = one clip per lane, lockstep = 1 lane)")
    p.add_argument("--batch", type=int, default=0, help="alias: --clips-per-gpu B --mode lockstep")
    p.add_argument("--steps-per-graph", type=int, default=0)
    p.add_argument("--config", default="zeggs", choices=["zeggs", "beat", "twh"],
                   help="zeggs = headline (BASELINE config[1]); beat/twh = DiffuseStyleGesture+ dims, 1830-frame clip (config[4])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-postprocess", action="store_true")
    p.add_argument("--cpu-baseline-steps", type=int, default=1000)
    p.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for a single rank")
    p.add_argument("--config3", default="auto", choices=["auto", "on", "off"],
                   help="second timed pass at config[3]'s per-GPU share (16 clips: 4 lanes x batch 4) reported as `config3` "
                        "(auto: whenever --gpus > 1 runs the default 1-clip-per-GPU workload)")
    p.add_argument("--sub-records", default="auto", choices=["auto", "on", "off"],
                   help="config2 / config3 / config4 / stream sub-records in the JSON line (auto: the default 1-GPU config[1] run)")
    a = p.parse_args(argv)
    if a.batch and not a.clips_per_gpu:
        a.clips_per_gpu, a.mode = a.batch, "lockstep"
    a.clips_per_gpu = a.clips_per_gpu or 1
    if a.lanes:
        if a.clips_per_gpu % a.lanes:
            p.error("--clips-per-gpu must be a multiple of --lanes")
        a.mode = "lanes" if 1 < a.lanes < a.clips_per_gpu else ("lockstep" if a.lanes == 1 else "streams")
    elif a.mode == "auto":
        # the command processor overlaps one queue per compute pipe: 4 lanes, the other clips ride in the lanes' batches
        # (DDIM-50: the step loops are 20x shorter and the per-window host work of 4 lanes shows -- one lock-step batch)
        a.lanes = 1 if a.sampler == "ddim50" else min(4, a.clips_per_gpu)
        while a.clips_per_gpu % a.lanes:
            a.lanes -= 1
        a.mode = "lanes" if 1 < a.lanes < a.clips_per_gpu else ("lockstep" if a.lanes == 1 else "streams")
    else:
        a.lanes = a.clips_per_gpu if a.mode == "streams" else 1
    return a


def self_launch(a):
    """`python bench.py --gpus N` with no launcher environment: create the N ranks (one per GPU) ourselves."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def host_cpu():
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "nproc": os.cpu_count() or 1}


def cpu_baseline(n_steps, one_core_steps=None):
    """Times the CPU oracle (numpy restatement of the reference path, validated against goldens) on a bounded sample of the
    config[0] workload: `n_steps` DDPM steps of one ZEGGS window at batch 1 (default: a whole 1000-step window, a quarter of
    a clip), on the best BLAS thread count found by a short sweep, plus a shorter single-thread run."""
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
    run = lambda k: sampler.p_sample_loop(d, m, shape, nf, {"y": y}, skip_timesteps=1000 - k)
    run(5)         # warm-up
    # BLAS thread count: the GEMMs of one step are small (89 x 256 x 1024 at most), so all host threads is not the
    # fastest setting; a short sweep picks the best one and `cores` reports the threads actually used
    cores = os.cpu_count() or 1
    one_core_steps = one_core_steps if one_core_steps is not None else max(10, min(300, n_steps // 3))
    one_core = None
    try:
        from threadpoolctl import threadpool_limits
        best = None
        for nt in sorted({1, 4, 8, 16, 32, cores}):
            if nt > cores:
                continue
            with threadpool_limits(limits=nt):
                t0 = time.perf_counter()
                run(min(20, n_steps))
                dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        cores = best[1]
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            run(one_core_steps)
            one_core = 1000.0 * (time.perf_counter() - t0) / one_core_steps
        with threadpool_limits(limits=cores):
            t0 = time.perf_counter()
            run(n_steps)
            dt = time.perf_counter() - t0
    except ImportError:
        t0 = time.perf_counter()
        run(n_steps)
        dt = time.perf_counter() - t0
    ms_step = 1000.0 * dt / n_steps
    fps = lambda ms: round(320.0 / (4000 * ms / 1000.0), 3)
    out = {"value": fps(ms_step), "unit": "frames/s", "cores": int(cores), "kind": "port",
           "ms_per_denoise_step": round(ms_step, 3), "host": host_cpu(),
           "sample": f"{n_steps} DDPM steps of one 88-frame ZEGGS window (batch 1, fp32 numpy oracle, {cores} BLAS threads); "
                     f"a 320-frame clip is 4 such windows of 1000 steps"}
    if one_core is not None:
        out["one_core"] = {"value": fps(one_core), "ms_per_denoise_step": round(one_core, 3), "steps": one_core_steps}
    return out


# algorithmic work per denoising step of ONE clip (SURVEY s8d / DESIGN.md s4): per-step weight parameters, fp32 state bytes
# (x_t in, noise in, x_{t-1} out) and GFLOP
ALGO = {"zeggs": (7.183e6, 1.205e6, 1.3152), "beat": (13.25e6, 3.694e6, 4.183), "twh": (20.23e6, 4.018e6, 6.311)}


def pmc_traffic(config, precision, B, kset):
    """HBM-side bytes per denoising step of ONE lane of batch B under kernel set `kset`, from the newest committed rocprofv3 PMC passes
    (tools/measure_traffic.sh -> tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH doubled as the MI355X guide
    prescribes for wide coalesced reads; HIP-launch path, the profiler cannot see AQL packets).  (corrected, raw, file) or None."""
    import glob
    if precision != "bf16":
        return None
    name = f"r*_traffic_{config}_b{B}_bf16.json" if B == 1 else f"r*_traffic_{config}_b{B}_{kset}_bf16.json"
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", name)))       # newest round last
    if not cands:
        return None
    d = json.load(open(cands[-1]))
    return d["traffic_bytes_per_step_fetch_x2"], d["traffic_bytes_per_step_raw"], os.path.basename(cands[-1])


def roofline_record(config, precision, NC, NL, B, us, with_traffic=False, kset=None):
    """The roofline object of one workload: `us` = time in which all NC clips in flight advance one denoising step."""
    wparams, sbytes, gflop = ALGO[config]
    wbytes = wparams * {"bf16": 2, "bf16w2": 4, "fp32": 4}[precision]      # (bf16w2: hi + lo bf16 per weight)
    abytes = wbytes + sbytes * NC           # the NC clips in flight share one pass over the weights
    if NC >= 8:
        t = pmc_traffic(config, precision, B, kset) if (with_traffic and kset) else None
        # from 8 clips in flight (>= 712 token rows) the path is a dense contraction -> MFMA roofline
        ach = gflop * NC / (us * 1e-6) / 1e3
        peak = 157.3 if precision == "fp32" else 2500.0
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 5),
                # HBM-side bytes per denoising step of all NC clips: the PMC passes of ONE lane of batch B (HIP-launch path) x NL lanes
                "traffic": None if t is None else t[0] * NL, "traffic_raw": None if t is None else t[1] * NL,
                "traffic_source": None if t is None else t[2] + (f" (one lane of batch {B}; x {NL} lanes)" if NL > 1 else ""),
                "algorithmic_bytes_per_denoise_step": abytes, "algorithmic_gflop_per_denoise_step": gflop * NC,
                "note": f"{NC} clips in flight ({NL} lane(s) x batch {B}): achieved = {gflop} GFLOP x {NC} clips / time in which all of them "
                        "advance one denoising step; peak = dense MFMA " + ("fp32" if precision == "fp32" else "bf16")}
    achieved = abytes / (us * 1e-6) / 1e9
    # HBM-side traffic per step from the rocprofv3 PMC passes (tools/pmc_traffic.py; FETCH_SIZE doubled as the MI355X guide
    # prescribes for wide coalesced reads), measured for the headline configuration only
    traffic, tsrc = None, None
    if with_traffic and NC == 1:
        t = pmc_traffic(config, precision, 1, None)
        if t is not None:
            traffic, tsrc = t[0], t[2]
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes_per_denoise_step": abytes,
            "note": "one denoising step = a chain of dependent kernel dispatches (2 + 3*L in the batch-1 latency set); achieved = "
                    "algorithmic bytes / time per step, timed from the first doorbell to the completion signal of "
                    "the last AQL packet (HIP events around the loop on the HIP-launch path); traffic = PMC bytes of "
                    "the committed rocprofv3 passes (separate runs), not of this run"}


class Workload:
    """One arrangement of clips on this rank's GPU: NC clips as NL sampling lanes x batch B, synthetic inputs resident in HBM.
    `one_pass(i)` samples every clip once (all windows); `step_us()` = device time in which all NC clips advanced one denoising
    step during the last pass."""

    def __init__(self, a, config, NC, NL, sampler, local, rank, world, library, emu, n_windows=None, precision=None):
        import torch
        from diffusestylegesture_amd import config as C
        from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
        from diffusestylegesture_amd.model import DSGDenoiser
        from diffusestylegesture_amd.parallel import shard_clips
        from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs
        self.config, self.NC, self.NL, self.B, self.sampler, self.emu = config, NC, NL, NC // NL, sampler, emu
        self.precision = precision or a.precision
        self.cfg = cfg = C.TINY if emu else C.CONFIGS[config]
        B = self.B
        # clip c -> rank c % world (parallel.shard_clips: the map gather_poses inverts): this rank's clips, dealt to its lanes in order
        my_clips = shard_clips(world * NC, rank, world)
        lane_clips = [my_clips[ln * B:(ln + 1) * B] for ln in range(NL)]
        self.model = DSGDenoiser(cfg, precision=self.precision, max_batch=B, device=local, steps_per_graph=a.steps_per_graph, library=library)
        self.model.load_state_dict(synth_state_dict(cfg, 20240))
        self.lanes = [self.model] + [self.model.clone() for _ in range(NL - 1)]
        self.diffusion = create_gaussian_diffusion("ddim50" if sampler == "ddim50" else "", library=library)
        self.sample_fn = self.diffusion.ddim_sample_loop if sampler == "ddim50" else self.diffusion.p_sample_loop
        self.skip = int(os.environ.get("DSG_BENCH_SKIP", "0")) if emu else 0
        if cfg.variant == 3:
            full_windows = 2 if emu else 4
            self.n_windows = n_windows or full_windows
            self.frames_per_clip = self.n_windows * cfg.stride                      # 320 nominal (312 emitted)
        else:
            full = 1830                                                             # BEAT-TWH sample.py:56, max_len=0
            full_windows = -(-full // cfg.stride)                                   # ceil -> 16 windows
            self.n_windows = n_windows or full_windows
            self.frames_per_clip = full if self.n_windows == full_windows else self.n_windows * cfg.stride
        # synthetic per-window audio features, resident in HBM before the clock starts; everything about a clip (features, seed
        # poses, its Philox stream = the id of the first clip of its lane + its position in the lane's batch) is a function of its id
        to_dev = (lambda x: x) if emu else (lambda x: torch.from_numpy(x).cuda(local))
        self.feats = [[to_dev(synth_window_inputs(cfg, B, window=w, clips=lane_clips[ln])["audio"]) for w in range(self.n_windows)]
                      for ln in range(NL)]
        self.seed0s = [to_dev(synth_window_inputs(cfg, B, window=0, clips=lane_clips[ln], seed_pose_scale=0.1)["seed"]) for ln in range(NL)]
        self.style = [1] + [0] * (cfg.style_dim_in - 1)
        if not emu:         # resident in HBM like the other inputs (a per-pass host -> device copy of it costs 1.8 ms per lane)
            self.style = torch.tensor([self.style] * B, dtype=torch.float32).cuda(local)
        self.lane_streams = [lc[0] for lc in lane_clips]
        self.n_denoise = self.diffusion.num_timesteps - self.skip

    def one_pass(self, i, skip=None):
        from diffusestylegesture_amd.sample import (generate_clip, generate_clip_dsgplus, generate_clips_streams,
                                                    generate_clips_streams_dsgplus)
        skip = self.skip if skip is None else skip
        ddim = self.sampler == "ddim50"
        if self.NL > 1 and self.cfg.variant == 3:
            return generate_clips_streams(self.lanes, self.diffusion, self.feats, self.style, seed=123456 + i, smoothing=True,
                                          skip_timesteps=skip, stream_ids=self.lane_streams, ddim=ddim)
        if self.NL > 1:
            return generate_clips_streams_dsgplus(self.lanes, self.diffusion, self.feats, self.style, self.seed0s, self.frames_per_clip,
                                                  seed=123456 + i, skip_timesteps=skip, stream_ids=self.lane_streams, ddim=ddim)
        if self.cfg.variant == 3:
            return generate_clip(self.model, self.diffusion, self.feats[0], self.style, seed=123456 + i, smoothing=True,
                                 sample_fn=self.sample_fn, stream_id=self.lane_streams[0], skip_timesteps=skip)
        return generate_clip_dsgplus(self.model, self.diffusion, self.feats[0], self.style, self.seed0s[0], self.frames_per_clip,
                                     seed=123456 + i, sample_fn=self.sample_fn, stream_id=self.lane_streams[0], skip_timesteps=skip)

    def step_us(self):
        if self.NL > 1:       # all lanes advance one step in: (slowest lane's time) / steps
            return max(1000.0 * ln.last_sample_ms()[0] / max(ln.last_sample_ms()[1], 1) for ln in self.lanes)
        return self.diffusion.last_step_time_us()

    def describe(self):
        arr = f"{self.NL} lane(s) x batch {self.B}" if self.NC > 1 else "batch 1"
        return (f"{self.NC} clip(s) in flight per GPU ({arr}), {self.frames_per_clip}-frame {self.config.upper()} clip "
                f"({self.n_windows} windows x {self.n_denoise} denoising steps), {self.sampler.upper()} {self.precision}")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import numpy as np
    import torch
    emu = os.environ.get("DSG_BENCH_EMU") == "1"          # TEST INFRASTRUCTURE ONLY (tests/test_bench_launch.py): CPU emulator, gloo
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not emu:
        torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if emu else "nccl")          # "nccl" = RCCL on ROCm
    from diffusestylegesture_amd import lib as L

    library = L.DSGLibrary(os.path.join(ROOT, "tests", "emu", "_build", "libdsg_emu.so")) if emu else None
    NC, NL = a.clips_per_gpu, a.lanes
    B = NC // NL
    wl = Workload(a, a.config, NC, NL, a.sampler, local, rank, world, library, emu)
    cfg, diffusion, model = wl.cfg, wl.diffusion, wl.model
    frames_per_clip, n_windows, skip = wl.frames_per_clip, wl.n_windows, wl.skip

    def sync():
        if not emu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not emu:
                torch.cuda.synchronize()

    def timed(w, passes, gather_total=0, warm_skip=None):
        """Warm-up pass, then `passes` passes between barrier + synchronize on both sides (+ the gather to rank 0, max over ranks)."""
        w.one_pass(0, skip=warm_skip)
        sync()
        t = time.perf_counter()
        us = []
        for i in range(passes):
            p = w.one_pass(1 + i)
            us.append(w.step_us())
        if dist is not None and gather_total:
            from diffusestylegesture_amd.parallel import gather_poses
            gather_poses(p, gather_total, dist, dst=0, device=None if emu else f"cuda:{local}")
        sync()
        dt = time.perf_counter() - t
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if emu else f"cuda:{local}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, float(np.mean(us))

    def sub_record(label, config, NC_, NL_, sampler, passes, n_windows=None, warm_skip=None, gather=False, precision=None):
        """One BASELINE configuration other than the headline's, timed like the main region on a bounded sample."""
        w = Workload(a, config, NC_, NL_, sampler, local, rank, world, library, emu, n_windows=n_windows, precision=precision)
        dt, us = timed(w, passes, gather_total=world * NC_ if gather else 0, warm_skip=warm_skip)
        rec = {"baseline_config": label, "workload": f"{world} GPU(s) x " + w.describe(), "clips": world * NC_,
               "value": round(world * NC_ * passes * w.frames_per_clip / dt, 2), "unit": "frames/s", "passes": passes,
               "ms_per_pass": round(1000.0 * dt / passes, 3), "us_per_denoise_step": round(us, 2),
               "kernel_set": w.lanes[0].last_kernel_set(), "sample_path": w.lanes[0].last_sample_path(),
               "roofline": roofline_record(config, w.precision, NC_, NL_, w.B, us, with_traffic=True, kset=w.lanes[0].last_kernel_set()) if us > 0 else None}
        if precision:
            rec["dtype"] = w.precision
        if NC_ > 1:
            rec["us_per_denoise_step_all_clips"] = rec["us_per_denoise_step"]
        del w
        return rec

    for i in range(a.warmup):
        wl.one_pass(i)
    sync()
    step_us = []
    t0 = time.perf_counter()
    poses = None
    for i in range(a.steps):
        poses = wl.one_pass(a.warmup + i)
        step_us.append(wl.step_us())
    gathered = poses
    if dist is not None:        # the only exchange of the path: finished poses -> rank 0 (RCCL over xGMI)
        from diffusestylegesture_amd.parallel import gather_poses
        gathered = gather_poses(poses, world * NC, dist, dst=0, device=None if emu else f"cuda:{local}")
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if emu else f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # ---- sub-records: the other BASELINE configurations, bounded (module docstring).  config3 is also what a multi-GPU run of the
    #      default workload carries (N x 16 clips gathered over RCCL); the single-GPU-only records run on one rank only.
    headline = cfg.variant == 3 and a.sampler == "ddpm" and NC == 1 and a.config == "zeggs"
    want_sub = a.sub_records == "on" or (a.sub_records == "auto" and headline and world == 1 and not emu)
    want_c3 = a.config3 == "on" or (a.config3 == "auto" and headline and (a.gpus > 1 or want_sub))
    subs = {}
    if want_c3 and cfg.variant == 3 and a.sampler == "ddpm":
        # "one clip per stream" is realised as 4 HSA queues x batch 4: four compute queues overlap on this GPU, a fifth collapses the
        # rate (16 x 1: 726 frames/s vs 5817 for 4 x 4; DESIGN.md s7, profiles/r04_b_multiproc_lanes.log)
        subs["config3"] = sub_record("config[3] per-GPU share: 16 clips, one clip per stream realised as 4 queues x batch 4 (a 5th "
                                     "queue collapses the rate)", "zeggs", 16, 4, "ddpm", 1, gather=True)
    if want_sub and world == 1:
        subs["config2"] = sub_record("config[2]: 50-step DDIM, batch 16 in lock step", "zeggs", 16, 1, "ddim50", 3)
        subs["config4"] = {
            name: sub_record(f"config[4]: DiffuseStyleGesture+ {name.upper()} denoiser, batch 1, 2 of the 16 windows of an 1830-frame clip",
                             name, 1, 1, "ddpm", 1, n_windows=2, warm_skip=900)
            for name in ("beat", "twh")}
        # ... and DSG+ with clips in flight (round 6: the ROWS kernel set at latent_dim 384 -- direct QKV + k_attn + k_ffn<OP> on 16-row tiles, streamed pose
        # embedding): 64 BEAT clips as 4 lanes x batch 16, 1 of the 16 windows
        subs["config4"]["beat_64clips"] = sub_record("DiffuseStyleGesture+ BEAT denoiser, 64 clips per GPU (4 lanes x batch 16), 1 of the 16 windows of an 1830-frame clip",
                                                     "beat", 64, 4, "ddpm", 1, n_windows=1, warm_skip=960)
        subs["stream"] = sub_record("256 clips per GPU (4 lanes x batch 64, STREAM kernel set)", "zeggs", 256, 4, "ddpm", 1, warm_skip=960)
        # config[1] in the other two arithmetic modes (tolerances against the reference .bvh: README "which precision"): fp32 is the
        # reference's own arithmetic (main/train/training_loop.py:39), bf16w2 keeps weights and the step's own GEMM operands as hi + lo bf16
        if a.precision == "bf16":
            subs["precision"] = {prec: sub_record(f"config[1] in {prec}: 1 clip, batch 1, 320-frame ZEGGS clip, 4 x 1000 DDPM steps", "zeggs", 1, 1, "ddpm", 1,
                                                  warm_skip=900, precision=prec) for prec in ("fp32", "bf16w2")}
            # ... and 16 clips in lock step in bf16w2 (round 6: the ROWS kernel set on two-register fragments), one window of 1000 steps
            subs["precision"]["bf16w2_16clips"] = sub_record("16 clips in lock step in bf16w2 (1 of the 4 windows of a 320-frame ZEGGS clip, 1000 DDPM steps)",
                                                             "zeggs", 16, 1, "ddpm", 1, n_windows=1, warm_skip=960, precision="bf16w2")
    if rank == 0 and os.environ.get("DSG_BENCH_DUMP"):      # TEST INFRASTRUCTURE (tests/test_bench_launch.py): the gathered poses, by clip id
        np.save(os.environ["DSG_BENCH_DUMP"], np.asarray(gathered, np.float32))
    if rank == 0:
        n_clips = world * NC * a.steps
        value = n_clips * frames_per_clip / dt
        emitted = int(poses.shape[1])
        n_denoise = wl.n_denoise
        us = float(np.mean(step_us))
        if not us > 0:      # no device timer (emulated test run): wall clock per denoising step
            us = 1e6 * dt / (a.steps * n_windows * n_denoise)
        roof = roofline_record(a.config, a.precision, NC, NL, B, us, with_traffic=True, kset=model.last_kernel_set())
        out = {
            "metric": (f"gesture frames/sec, {'1000-step DDPM' if a.sampler == 'ddpm' else '50-step DDIM'}, "
                       + ("320-frame ZEGGS clip" if a.config == "zeggs" else f"1830-frame {a.config.upper()} clip (DSG+)")
                       + (", 1/2/4/8 MI355X" if a.config == "zeggs" and a.sampler == "ddpm" else "")),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1000.0 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic" + (" (EMULATED ON CPU: test run, not a measurement)" if emu else ""),
            "config": {"workload": "1xMI355X per rank, " + wl.describe()
                                   + (" -- more than one clip per GPU: 'one clip per stream' is realised as <= 4 HSA queues x a batch per queue, "
                                      "because the 5th compute queue of a GPU collapses the rate" if NC > 1 and NL > 1 else ""),
                       "clips_per_gpu": NC, "lanes": NL, "batch_per_lane": B, "mode": a.mode if NC > 1 else "batch1",
                       "frames_nominal_per_clip": frames_per_clip,
                       "frames_emitted_per_clip": emitted, "denoise_steps_per_window": n_denoise, "parallelism": f"clips x{world}"},
            "value_emitted_frames": round(n_clips * emitted / dt, 2),
            "sample_path": diffusion.last_sample_path(),
            "kernel_set": model.last_kernel_set(),
            "collective_backend": (None if dist is None else ("gloo" if emu else "nccl")),
            "fence_free_packets": bool(model.last_sample_fence_free()),
            "us_per_denoise_step": round(us, 2),
            "roofline": roof,
        }
        if not a.no_postprocess and cfg.variant == 3 and not emu and gathered is not None:
            # de-normalisation + Savitzky-Golay + .bvh text for every clip of the last pass (C++, host threads), outside the
            # timed region as the metric defines it; value_end_to_end charges it to the job
            from diffusestylegesture_amd.bvh import pose2bvh_batch
            ms = np.load(os.path.join(ROOT, "diffusestylegesture_amd", "data", "zeggs_mean_std.npz"))
            g = np.ascontiguousarray(np.asarray(gathered, np.float32))
            with tempfile.TemporaryDirectory() as td:
                paths = [os.path.join(td, f"clip{c:03d}.bvh") for c in range(g.shape[0])]
                pose2bvh_batch(g, paths, smoothing=True, mean=ms["mean"], std=ms["std"])        # warm (thread start, page faults)
                t1 = time.perf_counter()
                pose2bvh_batch(g, paths, smoothing=True, mean=ms["mean"], std=ms["std"])
                post = time.perf_counter() - t1
                out["bvh_bytes_per_clip"] = os.path.getsize(paths[0])
            out["postprocess_ms_per_clip"] = round(1000.0 * post / g.shape[0], 3)
            out["postprocess_ms_per_pass"] = round(1000.0 * post, 3)
            out["value_end_to_end"] = round(n_clips * frames_per_clip / (dt + post * a.steps), 2)
        out.update(subs)
        if world == 1 and not a.no_cpu_baseline and a.config == "zeggs" and a.sampler == "ddpm" and not emu:
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
