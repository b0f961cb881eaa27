"""XCD-pinned lanes (or any other submission variant selected by --env) on the GPU: bit-identity with the fenced submission and
time per step.   python tools/pin_check.py [--lanes 1,8,16] [--skip 0] [--windows 2] [--env DSG_PIN=1]
Every lane's sample must equal the sample of the same (seed, stream) through DSG_PIN=0 exactly -- a stale read through a cache
the missing fences no longer invalidate shows up as a difference."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffusestylegesture_amd import config as C                      # noqa: E402
from diffusestylegesture_amd.diffusion import create_gaussian_diffusion      # noqa: E402
from diffusestylegesture_amd.model import DSGDenoiser                # noqa: E402
from diffusestylegesture_amd.synth import synth_state_dict, synth_window_inputs      # noqa: E402


BATCH = 1


def model(cfg, env):
    for kv in env.split(","):
        k, v = kv.split("=")
        os.environ[k] = v
    m = DSGDenoiser(cfg, precision="bf16", max_batch=BATCH)
    m.load_state_dict(synth_state_dict(cfg, 1))
    return m


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--lanes", default="1,8,16")
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--windows", type=int, default=2)
    p.add_argument("--config", default="zeggs")
    p.add_argument("--base-env", default="DSG_PIN=0,DSG_UC=0", help="environment of the reference handle (NAME=VALUE,...)")
    p.add_argument("--env", default="DSG_PIN=1", help="environment of the handle under test")
    p.add_argument("--batch", type=int, default=1, help="clips per lane")
    a = p.parse_args()
    global BATCH
    BATCH = a.batch
    cfg = C.CONFIGS[a.config]
    shape = (a.batch, cfg.njoints, 1, cfg.n_poses)
    d = create_gaussian_diffusion()
    nmax = max(int(x) for x in a.lanes.split(","))
    ys = [[{"y": synth_window_inputs(cfg, a.batch, window=w, clip0=i * a.batch, seed_pose_scale=0.2)} for i in range(nmax)] for w in range(a.windows)]
    m0 = model(cfg, a.base_env)
    want = [[d.manual_seed(100 + i, i).p_sample_loop(m0, shape, clip_denoised=False, model_kwargs=ys[w][i], skip_timesteps=a.skip)
             for i in range(nmax)] for w in range(a.windows)]
    print("fenced: path", m0.last_sample_path(), "%.2f us/step" % d.last_step_time_us(), flush=True)
    lanes0 = [m0] + [m0.clone() for _ in range(nmax - 1)]
    m = model(cfg, a.env)
    lanes = [m] + [m.clone() for _ in range(nmax - 1)]
    bad = 0
    for n in (int(x) for x in a.lanes.split(",")):
        if n > 1 and a.batch > 1:       # lanes pick their kernel set by lane count and batch: the reference runs the same arrangement
            for w in range(a.windows):
                ref = d.manual_seed(0, 0).p_sample_loop_multi(lanes0[:n], shape, ys[w][:n], seeds=[100 + i for i in range(n)],
                                                              stream_ids=list(range(n)), skip_timesteps=a.skip)
                for i in range(n):
                    want[w][i] = np.asarray(ref[i]).copy()
        for rep in range(2):
            for w in range(a.windows):
                t0 = time.perf_counter()
                if n == 1:
                    got = [d.manual_seed(100, 0).p_sample_loop(m, shape, clip_denoised=False, model_kwargs=ys[w][0], skip_timesteps=a.skip)]
                else:
                    got = d.manual_seed(0, 0).p_sample_loop_multi(lanes[:n], shape, ys[w][:n], seeds=[100 + i for i in range(n)],
                                                                  stream_ids=list(range(n)), skip_timesteps=a.skip)
                dt = time.perf_counter() - t0
                diff = [float(np.abs(np.asarray(got[i]) - np.asarray(want[w][i])).max()) for i in range(n)]
                nb = sum(x != 0.0 for x in diff)
                bad += nb
                us = d.last_step_time_us()
                print("lanes %2d rep %d window %d: path %s  %.2f us/step  (%.0f frames/s in the loop)  wall %.1f ms  mismatching lanes %d  max |diff| %.3g"
                      % (n, rep, w, m.last_sample_path(), us, n * cfg.n_poses / (us * (1000 - a.skip) * 1e-6), dt * 1e3, nb, max(diff)), flush=True)
    print("PIN CHECK", "OK" if bad == 0 else "FAILED (%d lane results differ)" % bad)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
