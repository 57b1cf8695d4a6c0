import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_product_library():
    """The in-tree C-ABI library is git-ignored: build it (hipcc cross-compiles gfx950 without a GPU, ~1 min) when a fresh
    checkout runs the tests before __graft_entry__.build()."""
    if not os.path.exists(os.path.join(ROOT, "synthesis_amd", "libsynthesis_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "synthesis_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (test infrastructure). Built on demand if the .so is missing."""
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def free_port():
    """A TCP port nobody listens on right now, for the rendezvous of a torch.distributed.run child (a fixed number collides with a
    previous run's socket in TIME_WAIT or with another job on the host)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def debug_shapes_built():
    """True when the library was built with `make DEBUG_SHAPES=1` (two-trees-per-lane and producer/consumer kernels: measured dead
    ends, forced-only launch shapes). The default library ships without them and their parity tests skip."""
    from synthesis_amd.engine import load_library

    return hasattr(load_library(), "syn_internal_debug_shapes")


@pytest.fixture
def debug_shapes():
    if not debug_shapes_built():
        pytest.skip("library built without DEBUG_SHAPES=1: the lane2 / producer-consumer debug kernels are not in it")
