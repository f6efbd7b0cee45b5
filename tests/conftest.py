import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The in-tree build normally travels with the snapshot; compile it if this checkout has none (hipcc
    cross-compiles gfx950 without a GPU). The product itself never falls back to anything: it needs the library."""
    lib = os.path.join(ROOT, "pathtrace-rs_amd", "_build", "libptgpu.so")
    host = os.path.join(ROOT, "pathtrace-rs_amd", "_build", "libpthost.so")
    if not (os.path.exists(lib) and os.path.exists(host)):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pathtrace-rs_amd"), "all"])


def load_ptgpu():
    """The product package directory is `pathtrace-rs_amd` (not an importable identifier)."""
    _ensure_built()
    name = "pathtrace_rs_amd_ptgpu"
    if name in sys.modules:
        return sys.modules[name]
    # torch ships its own copy of the HIP runtime; libptgpu.so uses /opt/rocm's. Both live in one process as long as torch's
    # initialises FIRST (the other way round torch reports "No HIP GPUs are available"), whatever order the tests run in.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "pathtrace-rs_amd", "ptgpu.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def ptgpu():
    return load_ptgpu()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding
