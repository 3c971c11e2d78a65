import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def _ensure_built():
    libs = ["raytracer_amd/librtgpu.so", "raytracer_amd/librthost.so", "oracle/liboracle.so"]
    if all(os.path.exists(os.path.join(ROOT, l)) for l in libs):
        return
    if os.path.isdir("/opt/rocm") and os.path.exists(os.path.join(ROOT, "Makefile")):
        subprocess.run(["make", "-C", ROOT, "-j4"], check=True)


_ensure_built()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx():
    import raytracer_amd as ra
    ctx = ra.Context(0)   # raises when there is no device: GPU tests must not pass on a fallback
    yield ctx
    ctx.close()
