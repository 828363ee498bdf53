import importlib
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PKG_NAME = "opencv-simpleslam_amd"

if os.environ.get("SSLAM_TEST_CV2_CLASSES") == "1":
    # a child run of tests/test_cv2_classes.py: `cv2` (absent from the image) is the stand-in of tests/cv2_stub.py with
    # KeyPoint / DMatch / KeyPoint_convert from the C module - installed BEFORE the product binds cv2 at import, so the
    # overlay takes the branch it takes wherever the reference really runs (slam/core/types.py: HAVE_CV2)
    sys.path.insert(0, str(ROOT / "tests"))
    import cv2_stub
    cv2_stub.install(native_classes=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Bring libsslam_hip.so up to date with the sources before any test runs, the same way
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  The build is
    incremental and keyed on source content + flags (build.py), so an up-to-date library costs a
    few milliseconds and a stale one (edited source, leftover experiment flags) is never tested."""
    import importlib.util
    import shutil
    if shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists():
        return                                   # the C-ABI tests will say so loudly
    spec = importlib.util.spec_from_file_location("sslam_build", ROOT / PKG_NAME / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_native(force=False, verbose=False)


def load_pkg(sub: str = ""):
    """Import (a submodule of) the hyphen-named product package."""
    return importlib.import_module(PKG_NAME + (("." + sub) if sub else ""))


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def native():
    return load_pkg("_native")


@pytest.fixture(scope="session")
def gpu_ctx(native):
    """Process-wide GPU context; fails (not skips) when the HIP library or the
    device is missing, so a silent fallback can never make GPU tests pass."""
    assert native.device_count() >= 1, "no HIP device visible - GPU tests need an MI355X"
    return native.default_context(0)
