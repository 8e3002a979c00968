import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The HIP library is git-ignored: on a clean tree (or after a source change) build it before any test loads it.
    hipcc cross-compiles gfx950 without a GPU; on the GPU box the prebuilt in-tree .so travels with the snapshot."""
    import subprocess
    lib = os.path.join(ROOT, "go-kzg_amd", "libkzg_hip.so")
    src_dir = os.path.join(ROOT, "go-kzg_amd", "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".hpp"))] + [os.path.join(ROOT, "include", "kzg_hip.h")]
    stale = not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs)
    if stale and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-C", src_dir, "-j4"], stdout=subprocess.DEVNULL)
    return lib
