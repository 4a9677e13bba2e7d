import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch ships its own HIP runtime and libslimm_hip.so links the system one; whichever loads first serves both (same
    # SONAME).  Load torch's first, as bench.py and smoke() do, so that tests using torch tensors next to the library do
    # not depend on which test happened to run first.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """SLIMM_EMU=1 (developer aid, off by default): the kernels' logic on the CPU.  tests/native/libslimm_emu.so is the
    product's .hip sources compiled with g++ against a host stand-in for the HIP runtime (tests/native/hip_emu); with
    the variable set, the `gpu` tests that need no torch device tensor run against it here.  The driver's runs
    (`-m "not gpu"` here, `-m gpu` on the MI355X) never set it, and slimm_amd/ knows nothing about it."""
    if os.environ.get("SLIMM_EMU") != "1":
        return
    import subprocess

    from slimm_amd import capi

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "native"), "-j8"])
    capi.LIB_PATH = os.path.join(ROOT, "tests", "native", "libslimm_emu.so")
    capi._lib = None
    import inspect

    skip = pytest.mark.skip(reason="needs a real GPU (torch device tensors / the command line binary)")
    for it in items:
        if "gpu" not in it.keywords:
            continue
        try:
            src = inspect.getsource(it.function)
        except (OSError, TypeError):
            src = ""
        if any(k in src for k in ("torch", "tensor", "sharded_profile", "force_exchange")) or "full_size" in it.name or "test_cli_gpu" in it.nodeid or "run_slimm" in src or "SLIMM_BIN" in src or "big_configs" in it.name:
            it.add_marker(skip)


def gpu_available() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
