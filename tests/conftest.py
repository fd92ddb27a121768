import os
import sys

import pytest

# (before torch is imported) OpenMP pools sized for the host's 256 logical CPUs run into the container's CPU quota on the GPU
# boxes, which then stalls every thread of the process; see bench.py
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("MKL_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The tests bind the in-tree libr3dg_hip.so.  A fresh checkout has none (built artefacts are git-ignored): build it
    once per session when hipcc is there (it cross-compiles gfx950 without a GPU).  Building is not a fallback -- with
    no library and no compiler the tests that need it fail loudly."""
    from relightable3dgaussian_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH) and os.path.exists(build.HIPCC):
        build.build(verbose=False)


@pytest.fixture(scope="session")
def hip_lib():
    from relightable3dgaussian_amd import _lib
    return _lib.lib()
