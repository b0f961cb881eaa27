"""MI355X-native DDPM/DDIM sampling path of DiffuseStyleGesture (hand-written gfx950 kernels behind a C ABI)."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every dependent launch of the step loop starts
# by fetching ~300 B of kernargs, and from host memory that alone costs about 1 us per launch (measured: 344 -> 261 us
# per denoising step, profiles/r01_b_*).  Must be set before the HIP runtime initialises.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .config import DSGConfig, ZEGGS, BEAT, TWH, CONFIGS  # noqa: E402,F401

__all__ = ["DSGConfig", "ZEGGS", "BEAT", "TWH", "CONFIGS"]
