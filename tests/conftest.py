import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, exactly as __graft_entry__.build() does
    (hipcc cross-compiles gfx950 without a GPU).  Nothing happens when they are already there and newer than the sources."""
    lib = os.path.join(ROOT, "ucoslam-cv3_amd", "libucoslam_hip.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def hip_ctx():
    """A uh_ctx on cuda:0 bound to torch's current stream. GPU tests only."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch sees no GPU: the HIP path must run, there is no fallback")
    import ucoslam_cv3_amd as u

    torch.cuda.set_device(0)
    ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()
