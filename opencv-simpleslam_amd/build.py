"""Build libsslam_hip.so (gfx950) in-tree with hipcc.

`python opencv-simpleslam_amd/build.py` or `build_native()` from Python.  The
built library lands in opencv-simpleslam_amd/lib/ (git-ignored, travels to the
GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libsslam_hip.so"
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17",
    "-Wall", "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; this package builds only with the ROCm toolchain")
    return exe


def _digest(*parts) -> str:
    h = hashlib.sha256()
    for p in parts:
        h.update(p if isinstance(p, bytes) else str(p).encode())
        h.update(b"\0")
    return h.hexdigest()


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """Incremental build keyed on CONTENT, not mtimes: an object is reused only if its key file
    holds the hash of (compiler, flags, extra flags, the source, every header).  So an experiment
    build (SSLAM_EXTRA_HIPCC_FLAGS=-DSSLAM_DBG_NOMFMA=1 ...) can never survive into a default build,
    a source edit always rebuilds, and a snapshot copy that scrambles mtimes rebuilds nothing."""
    hipcc = _hipcc()
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + [PKG_DIR.parent / "include" / "sslam_hip.h"]
    sources = sorted(CSRC.glob("*.hip"))
    if not sources:
        raise RuntimeError(f"no .hip sources under {CSRC}")

    extra = os.environ.get("SSLAM_EXTRA_HIPCC_FLAGS", "").split()     # experiments only (e.g. -DSSLAM_DBG=1)
    base_key = _digest(hipcc, *HIPCC_FLAGS, "|", *extra, *[h.read_bytes() for h in headers])

    def compile_one(src: Path):
        obj = obj_dir / (src.stem + ".o")
        keyf = obj_dir / (src.stem + ".key")
        key = _digest(base_key, src.read_bytes())
        if force or not obj.exists() or not keyf.exists() or keyf.read_text().strip() != key:
            keyf.unlink(missing_ok=True)
            cmd = [hipcc, *HIPCC_FLAGS, *extra, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
            if verbose and res.stderr.strip():
                print(res.stderr)
            keyf.write_text(key + "\n")
        return obj, key

    with ThreadPoolExecutor(max_workers=min(4, len(sources))) as ex:
        built = list(ex.map(compile_one, sources))
    objs = [o for o, _ in built]

    lib_keyf = LIB_DIR / "libsslam_hip.key"
    lib_key = _digest(*[k for _, k in built])
    if force or not LIB_PATH.exists() or not lib_keyf.exists() or lib_keyf.read_text().strip() != lib_key:
        lib_keyf.unlink(missing_ok=True)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        lib_keyf.write_text(lib_key + "\n")
    return LIB_PATH


if __name__ == "__main__":
    p = build_native(force="--force" in sys.argv, verbose=True)
    print("built", p)
