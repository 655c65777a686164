import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A clean checkout has no built libraries (they are git-ignored): build them once, exactly as
    ``__graft_entry__.build()`` does (hipcc cross-compiles for gfx950 without a GPU)."""
    import subprocess

    lib = os.path.join(ROOT, "magnify_amd", "_lib", "libmagnify_hip.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "magnify_amd", "csrc"), "-j4"], check=True,
                       stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libref_port.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load
