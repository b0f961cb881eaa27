"""Denoiser / sampler dimension sets for the DiffuseStyleGesture hot path.

Each `DSGConfig` names the dimensions the reference hard-codes:

* ZEGGS (DiffuseStyleGesture): `main/mydiffusion_zeggs/sample.py:51-56`
  (njoints=1141, latent_dim=256, n_seed=8, cond_mode cross_local_attention3_style1),
  `main/mydiffusion_zeggs/configs/DiffuseStyleGesture.yml:10` (n_poses=88),
  `main/model/mdm.py:51` (audio latent 64), `:131-140` (window 11, 8 local heads).
* BEAT / TWH (DiffuseStyleGesture+): `BEAT-TWH-main/mydiffusion_beat_twh/configs/DiffuseStyleGesture.yml`
  and `BEAT-TWH-main/model/mdm.py:83-118` (cross_local_attention4, window 15).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

VARIANT_DSG = 3       # cross_local_attention3_style1 (ZEGGS)
VARIANT_DSGPLUS = 4   # cross_local_attention4_style1 (BEAT / TWH)
VARIANT_DSGPP = 5     # cross_local_attention5_style1 (DiffuseStyleGesture++: + y['seed_last'], BEAT-TWH mdm.py:226-264)


@dataclass(frozen=True)
class DSGConfig:
    name: str
    variant: int          # 3, 4 or 5
    njoints: int          # J: pose feature dim
    n_poses: int          # T: frames per denoised window
    n_seed: int           # S: seed frames
    latent_dim: int       # D
    audio_src_dim: int    # A_src
    audio_dim: int        # A: audio latent
    style_dim_in: int     # one-hot style/speaker width
    window: int           # local attention window
    num_layers: int = 8
    num_heads: int = 4        # self-attention heads
    ff_size: int = 1024
    local_heads: int = 8      # mdm.py:58 self.num_head = 8
    pe_max_len: int = 5000    # mdm.py:373

    @property
    def audio_frames(self) -> int:
        # DSG: audio covers all T frames; DSG+ (attention4): T - S audio frames, the
        # first S "audio" rows are the per-frame seed embedding (BEAT-TWH mdm.py:188-190)
        # DSG++ (attention5): the last S rows are the embedding of y['seed_last'] as well (mdm.py:227-230; the caller
        # drops the last S audio frames, BEAT-TWH sample.py:104)
        if self.variant == VARIANT_DSG:
            return self.n_poses
        return self.n_poses - self.n_seed * (2 if self.variant == VARIANT_DSGPP else 1)

    @property
    def stride(self) -> int:
        return self.n_poses - self.n_seed

    @property
    def tok_style_dim(self) -> int:
        # width of embed_style output
        return 64 if self.variant == VARIANT_DSG else self.latent_dim

    def as_dict(self):
        return asdict(self)


ZEGGS = DSGConfig("zeggs", VARIANT_DSG, njoints=1141, n_poses=88, n_seed=8, latent_dim=256,
                  audio_src_dim=1024, audio_dim=64, style_dim_in=6, window=11)
BEAT = DSGConfig("beat", VARIANT_DSGPLUS, njoints=2052, n_poses=150, n_seed=30, latent_dim=384,
                 audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)
TWH = DSGConfig("twh", VARIANT_DSGPLUS, njoints=2232, n_poses=150, n_seed=30, latent_dim=512,
                audio_src_dim=1435, audio_dim=128, style_dim_in=17, window=15)
# tiny dims for fast CPU tests (same structure, small sizes; J deliberately odd)
# (the reference MDM classes hard-code 8 local heads, the window, and -- for ZEGGS -- the
#  1024->64 audio map and the 6->64 style map, so the tiny sets keep those)
TINY = DSGConfig("tiny", VARIANT_DSG, njoints=37, n_poses=22, n_seed=4, latent_dim=128,
                 audio_src_dim=1024, audio_dim=64, style_dim_in=6, window=11,
                 num_layers=2, num_heads=4, ff_size=128)
TINY4 = DSGConfig("tiny4", VARIANT_DSGPLUS, njoints=37, n_poses=30, n_seed=6, latent_dim=64,
                  audio_src_dim=40, audio_dim=16, style_dim_in=3, window=15,
                  num_layers=2, num_heads=2, ff_size=128)

TINY5 = DSGConfig("tiny5", VARIANT_DSGPP, njoints=37, n_poses=30, n_seed=6, latent_dim=64,
                  audio_src_dim=40, audio_dim=16, style_dim_in=3, window=15,
                  num_layers=2, num_heads=2, ff_size=128)
BEATPP = DSGConfig("beatpp", VARIANT_DSGPP, njoints=2052, n_poses=150, n_seed=30, latent_dim=384,
                   audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)

# BEAT-TWH-main's "DiffuseStyleGesture" (cross_local_attention3_style1_sample, BEAT-TWH-main/model/mdm.py:147-185): the ZEGGS
# conditioning scheme at BEAT dims -- window 15, audio covers all T frames (sample.py:100-102, :132-134)
BEAT3 = DSGConfig("beat3", VARIANT_DSG, njoints=2052, n_poses=150, n_seed=30, latent_dim=384,
                  audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)
TINY3B = DSGConfig("tiny3b", VARIANT_DSG, njoints=37, n_poses=30, n_seed=6, latent_dim=384,
                   audio_src_dim=40, audio_dim=16, style_dim_in=3, window=15,
                   num_layers=2, num_heads=6, ff_size=128)

# the remaining name x dataset pairs BEAT-TWH-main/mydiffusion_beat_twh/sample.py:299-323 accepts: TWH dims under
# "DiffuseStyleGesture" (attention3) and "DiffuseStyleGesture++" (attention5), and BEAT "v2" (njoints = motion_dim = 1141, the
# whole feature vector is kept: motion_feature_division = 1, sample.py:173-176)
TWH3 = DSGConfig("twh3", VARIANT_DSG, njoints=2232, n_poses=150, n_seed=30, latent_dim=512,
                 audio_src_dim=1435, audio_dim=128, style_dim_in=17, window=15)
TWHPP = DSGConfig("twhpp", VARIANT_DSGPP, njoints=2232, n_poses=150, n_seed=30, latent_dim=512,
                  audio_src_dim=1435, audio_dim=128, style_dim_in=17, window=15)
BEATV2 = DSGConfig("beatv2", VARIANT_DSGPLUS, njoints=1141, n_poses=150, n_seed=30, latent_dim=384,
                   audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)
BEATV2_3 = DSGConfig("beatv2_3", VARIANT_DSG, njoints=1141, n_poses=150, n_seed=30, latent_dim=384,
                     audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)
BEATV2PP = DSGConfig("beatv2pp", VARIANT_DSGPP, njoints=1141, n_poses=150, n_seed=30, latent_dim=384,
                     audio_src_dim=1434, audio_dim=96, style_dim_in=2, window=15)

CONFIGS = {c.name: c for c in (ZEGGS, BEAT, TWH, TINY, TINY4, TINY5, BEATPP, BEAT3, TINY3B, TWH3, TWHPP, BEATV2, BEATV2_3, BEATV2PP)}
# (name of the reference's yml, dataset, version) -> dims, as sample.py:297-323 sets them
DSGPLUS_CONFIGS = {
    ("DiffuseStyleGesture", "BEAT", "v0"): BEAT3, ("DiffuseStyleGesture+", "BEAT", "v0"): BEAT, ("DiffuseStyleGesture++", "BEAT", "v0"): BEATPP,
    ("DiffuseStyleGesture", "BEAT", "v2"): BEATV2_3, ("DiffuseStyleGesture+", "BEAT", "v2"): BEATV2, ("DiffuseStyleGesture++", "BEAT", "v2"): BEATV2PP,
    ("DiffuseStyleGesture", "TWH", "v0"): TWH3, ("DiffuseStyleGesture+", "TWH", "v0"): TWH, ("DiffuseStyleGesture++", "TWH", "v0"): TWHPP,
}
