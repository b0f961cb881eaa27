"""CPU, world_size 2, gloo: the N > 1 path (clip sharding + the single final gather).  Each rank samples its clips
with the emulated product library (test infrastructure) at tiny dims; rank 0 must end up with exactly what one process
computes for all clips, in clip order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusestylegesture_amd import config as C
from diffusestylegesture_amd.parallel import gather_poses, shard_clips
from tests.conftest import EMU_LIB


def _clip(model, diffusion, clip):
    from diffusestylegesture_amd.sample import generate_clip
    from diffusestylegesture_amd.synth import synth_window_inputs
    cfg = model.cfg
    feats = [synth_window_inputs(cfg, 1, window=w, clip0=clip)["audio"] for w in range(2)]
    return generate_clip(model, diffusion, feats, [1, 0, 0, 0, 0, 0], seed=7, stream_id=clip, skip_timesteps=997)[0]


def _make():
    from diffusestylegesture_amd import lib as L
    from diffusestylegesture_amd.diffusion import create_gaussian_diffusion
    from diffusestylegesture_amd.model import DSGDenoiser
    from diffusestylegesture_amd.synth import synth_state_dict
    lib = L.DSGLibrary(EMU_LIB)
    m = DSGDenoiser(C.TINY, precision="fp32", max_batch=1, library=lib)
    m.load_state_dict(synth_state_dict(C.TINY, 3))
    return m, create_gaussian_diffusion(library=lib)


def _worker(rank, world, port, n_clips, q):
    os.environ["DSG_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    m, d = _make()
    mine = np.stack([_clip(m, d, c) for c in shard_clips(n_clips, rank, world)])
    dist.barrier()
    out = gather_poses(mine, n_clips, dist, dst=0)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_assignment():
    assert shard_clips(5, 0, 2) == [0, 2, 4] and shard_clips(5, 1, 2) == [1, 3]
    assert sorted(sum((shard_clips(128, r, 8) for r in range(8)), [])) == list(range(128))


def test_two_rank_gather_matches_single_process(emu_lib):
    n_clips, world = 3, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, d = _make()
    ref = np.stack([_clip(m, d, c) for c in range(n_clips)])
    assert got.shape == ref.shape == (3, 2 * C.TINY.stride - C.TINY.n_seed, C.TINY.njoints)
    assert np.array_equal(got, ref)
