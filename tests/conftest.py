import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The eight-PROCESS cfg4 test (opt-in since round 4: SFGPU_CFG4_PROCS=1 -- nine processes on ONE device took 82, 460 and 863 s
    on three boxes) runs FIRST when it runs: on a device that earlier tests of the session have used it is slower still."""
    first = [it for it in items if it.name.startswith("test_cfg4_eight_ranks_share_the_gpu")]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest


@pytest.fixture(scope="session")
def built():
    """Everything compiled (libsfgpu.so for gfx950, the C oracle, oracle/_ref when possible)."""
    import __graft_entry__ as g
    so = os.path.join(ROOT, "sailfish_amd", "csrc", "libsfgpu.so")
    if not os.path.exists(so) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        g.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the HIP path has no CPU fallback")
    return torch.device("cuda:0")
