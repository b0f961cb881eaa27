import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libdsg_emu.so")
HIP_LIB = os.path.join(ROOT, "diffusestylegesture_amd", "csrc", "libdsg_hip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _blas_threads():
    """The numpy oracle multiplies small matrices (89 x 256 x 1024 at most): on a 256-thread host the default BLAS pool is ~20x
    SLOWER than 8 threads (bench.py's cpu_baseline sweep finds the same), which turned the oracle-backed GPU tests into a
    17-minute run.  Every test runs under an 8-thread limit."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits=8):
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _make(target, path):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ROOT, target])
    return path


@pytest.fixture(scope="session")
def hip_lib_path():
    """The product library (hipcc cross-compiles gfx950 without a GPU; it loads on a CPU-only host)."""
    return _make("all", HIP_LIB)


@pytest.fixture(scope="session")
def emu_lib():
    """TEST INFRASTRUCTURE: the product sources compiled for the host under the SIMT emulator (tests/emu)."""
    from diffusestylegesture_amd import lib as L
    return L.DSGLibrary(_make("emu", EMU_LIB))
