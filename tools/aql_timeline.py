#!/usr/bin/env python3
"""Per-packet timeline of the step loop ON THE PATH THAT IS TIMED (hand-written AQL packets, fence-free): start / end of every
dispatch of N consecutive denoising steps from the command processor's own timestamps (queue profiling mode + one completion
signal per traced packet; csrc/dsg_aql.h: Trace) -- rocprofv3 only sees the HIP-launch path.

    python tools/aql_timeline.py [--batch 1] [--kset auto] [--steps 600] [--first 200] [--n 64] [--out profiles/r03_aql_step_timeline.json]

Writes, per packet position of a step (mean over the traced steps): kernel, busy = end - start, gap = start - end of the
previous packet; their sums; the step time of the traced steps (start of step s+1 - start of step s); and the step time of an
UN-traced run of the same call for comparison (what bench.py reports as us_per_denoise_step)."""
import argparse
import ctypes as C
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch

from diffusestylegesture_amd import config as CFG
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
from diffusestylegesture_amd.model import DSGDenoiser
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=1)
p.add_argument("--kset", default="auto")
p.add_argument("--steps", type=int, default=600)
p.add_argument("--first", type=int, default=200)
p.add_argument("--n", type=int, default=32)
p.add_argument("--config", default="zeggs")
p.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16w2"])
p.add_argument("--out", default="")
p.add_argument("--lib", default="stamps",
               help="stamps: libdsg_hip_stamps.so (`make stamps`): in-kernel first-wave / last-wave stamps, ~1 %% overhead; "
                    "product: libdsg_hip.so with command-processor dispatch timestamps (queue profiling), ~20 %% overhead")
a = p.parse_args()
if a.lib != "product":      # (dev: `make dev DEVFLAGS=-DDSG_STAMPS=2`, the bf16-only development build with marks)
    # marks: `make marks` -- the stamps build + DSG_TL_MARK phase marks inside the kernels (dsg_kernels.h)
    os.environ["DSG_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusestylegesture_amd", "csrc", f"libdsg_hip_{a.lib}.so")
cfg = CFG.CONFIGS[a.config]
m = DSGDenoiser(cfg, precision=a.precision, max_batch=a.batch, device=0).set_kernel_set(a.kset)
m.load_state_dict(synth_state_dict(cfg, 20240))
d = create_gaussian_diffusion()
shape = (a.batch, cfg.njoints, 1, cfg.n_poses)
y = {"y": {k: torch.from_numpy(v).cuda() for k, v in synth_window_inputs(cfg, a.batch, window=1, seed_pose_scale=0.1).items()}}
skip = 1000 - a.steps
run = lambda: d.manual_seed(1, 0).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=y, skip_timesteps=skip)
run()
untraced = []
for _ in range(3):
    ref = run()
    untraced.append(d.last_step_time_us())
cdll = m.lib.cdll
cdll.dsg_debug_trace_arm.argtypes = [C.c_void_p, C.c_int, C.c_int]
cdll.dsg_debug_trace_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
m.lib.check(cdll.dsg_debug_trace_arm(m.handle, a.first, a.n))
out = run()
traced_run_us = d.last_step_time_us()
assert m.last_sample_path() == "aql"
buf = np.zeros(a.n * 64 * 2, np.float64)
ns, npk = C.c_int(), C.c_int()
names = C.create_string_buffer(1 << 16)
m.lib.check(cdll.dsg_debug_trace_get(m.handle, buf.ctypes.data, buf.size, C.byref(ns), C.byref(npk), names, len(names)))
S, L = ns.value, npk.value
t = buf[: S * L * 2].reshape(S, L, 2)
mangled = [n for n in names.value.decode().split(";") if n]


def short_name(m):          # _ZN3dsg6k_gemmINS_5PBF16ELi1ELi1ELi4ELi1ELi1EEEvNS_8GemmArgsE -> k_gemm<PBF16,1,1,4,1,1>
    mm = re.match(r"_ZN3dsg(\d+)", m)
    if not mm:
        return m
    n = int(mm.group(1))
    name = m[mm.end(): mm.end() + n]
    rest = m[mm.end() + n:]
    targs = []
    if rest.startswith("I"):
        body = rest[1: rest.index("EEv") + 1] if "EEv" in rest else rest
        targs = [t[0] or t[1] for t in re.findall(r"Li(\d+)E|NS_\d+(P[A-Z0-9]+)E", body)]
    return name + ("<" + ",".join(targs) + ">" if targs else "")


short = [short_name(n) for n in mangled]
marks = None
if a.lib == "marks" or a.lib.startswith("dev"):
    mb = np.zeros(S * L * 12 * 3, np.float64)
    nm = C.c_int()
    cdll.dsg_debug_trace_marks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    if cdll.dsg_debug_trace_marks(m.handle, mb.ctypes.data, mb.size, C.byref(nm)) == 0:      # (a dev build without -DDSG_STAMPS=2 has none)
        marks = mb.reshape(S, L, 12, 3)
busy = t[:, :, 1] - t[:, :, 0]
flat = t.reshape(S * L, 2)
gap = np.concatenate([[np.nan], flat[1:, 0] - flat[:-1, 1]]).reshape(S, L)      # gap BEFORE each packet (first traced packet: unknown)
step_time = (t[1:, 0, 0] - t[:-1, 0, 0])
pk = []
for i in range(L):
    pk.append({"packet": i, "kernel": short[i], "busy_us": round(float(busy[:, i].mean()), 3), "busy_us_min": round(float(busy[:, i].min()), 3),
               "gap_before_us": round(float(np.nanmean(gap[:, i])), 3)})
busy_sum, gap_sum = float(busy.mean(0).sum()), float(np.nanmean(gap, 0).sum())
by_kernel = {}
for q in pk:
    e = by_kernel.setdefault(q["kernel"], {"launches_per_step": 0, "busy_us": 0.0, "gap_before_us": 0.0})
    e["launches_per_step"] += 1; e["busy_us"] = round(e["busy_us"] + q["busy_us"], 3); e["gap_before_us"] = round(e["gap_before_us"] + q["gap_before_us"], 3)
res = {
    "what": "per-packet timeline of one denoising step on the fence-free AQL path: " +
            ("in-kernel first-wave-start / last-wave-end stamps (timeline build)" if a.lib != "product" else "command-processor dispatch timestamps (queue profiling: start = packet taken up, so the gaps read 0 and busy includes the dispatch overhead)"),
    "library": a.lib,
    "config": a.config, "batch": a.batch, "kernel_set": m.last_kernel_set(), "fence_free": bool(m.last_sample_fence_free()),
    "traced_steps": S, "packets_per_step": L,
    "us_per_step_untraced_runs": [round(v, 3) for v in untraced],
    "us_per_step_traced_run_whole_call": round(traced_run_us, 3),
    "us_per_step_of_traced_steps": round(float(step_time.mean()), 3),
    "sum_busy_us": round(busy_sum, 3), "sum_gaps_us": round(gap_sum, 3), "sum_busy_plus_gaps_us": round(busy_sum + gap_sum, 3),
    "ratio_traced_to_untraced": round((busy_sum + gap_sum) / min(untraced), 4),
    # what tracing adds sits at the END of every traced kernel: a wave cannot retire before its end-stamp store is acknowledged,
    # the stamp itself is taken before -- so `busy` is unperturbed and the measured gaps carry that tail
    "stamp_tail_us_per_packet": round((busy_sum + gap_sum - min(untraced)) / L, 3),
    "untraced_step_decomposition": {"us_per_step": round(min(untraced), 3), "kernels_busy_us": round(busy_sum, 3),
                                    "kernels_busy_share": round(busy_sum / min(untraced), 4),
                                    "boundaries_us": round(min(untraced) - busy_sum, 3), "per_boundary_us": round((min(untraced) - busy_sum) / L, 3)},
    "samples_identical_to_untraced": bool(np.array_equal(np.asarray(out.cpu()), np.asarray(ref.cpu()))),
    "by_kernel": by_kernel, "packets": pk,
}
if marks is not None:
    # per kernel (all its packets of a step, all traced steps): mark k -> waves that passed it per launch, mean and last-wave time in us
    # after the kernel's first wave start; `busy` of the same launches for scale.  What each mark means: the DSG_TL_MARK sites in csrc/.
    ph = {}
    for i in range(L):
        e = ph.setdefault(short[i], {"launches": 0, "busy_us": 0.0, "marks": {}})
        e["launches"] += 1; e["busy_us"] += float(busy[:, i].mean())
        for k in range(12):
            cnt = marks[:, i, k, 0]
            if cnt.max() <= 0:
                continue
            q = e["marks"].setdefault(k, {"waves": 0.0, "mean_us": 0.0, "last_wave_us": 0.0, "n": 0})
            q["waves"] += float(cnt.mean()); q["mean_us"] += float(marks[:, i, k, 1].mean()); q["last_wave_us"] += float(marks[:, i, k, 2].mean()); q["n"] += 1
    for e in ph.values():
        e["busy_us"] = round(e["busy_us"] / e["launches"], 3)
        for q in e["marks"].values():
            n = q.pop("n")
            for kk in ("waves", "mean_us", "last_wave_us"):
                q[kk] = round(q[kk] / n, 3)
    res["phase_marks"] = ph
print(json.dumps({k: v for k, v in res.items() if k != "packets"}, indent=1))
if marks is not None:
    for kn, e in res["phase_marks"].items():
        print(f"{kn[:70]:70s} busy {e['busy_us']:7.2f} us/launch  marks (mean / last wave, us after first wave start): " +
              "  ".join(f"[{k}] {q['mean_us']:.2f}/{q['last_wave_us']:.2f}" for k, q in sorted(e["marks"].items())))
for q in pk:
    print(f"{q['packet']:3d} {q['kernel'][:60]:60s} busy {q['busy_us']:7.2f} (min {q['busy_us_min']:6.2f})  gap before {q['gap_before_us']:6.2f}")
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
