"""Build libsslam_hip.so (gfx950) in-tree with hipcc.

`python opencv-simpleslam_amd/build.py` or `build_native()` from Python.  The
built library lands in opencv-simpleslam_amd/lib/ (git-ignored, travels to the
GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

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


def _needs_rebuild(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + [PKG_DIR.parent / "include" / "sslam_hip.h"]
    sources = sorted(CSRC.glob("*.hip"))
    if not sources:
        raise RuntimeError(f"no .hip sources under {CSRC}")

    extra = os.environ.get("SSLAM_EXTRA_HIPCC_FLAGS", "").split()     # experiments only (e.g. -DSSLAM_DBG=1)

    def compile_one(src: Path):
        obj = obj_dir / (src.stem + ".o")
        if force or extra or _needs_rebuild(obj, [src, *headers]):
            cmd = [hipcc, *HIPCC_FLAGS, *extra, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
            if verbose and res.stderr.strip():
                print(res.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources))

    if force or _needs_rebuild(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    p = build_native(force="--force" in sys.argv, verbose=True)
    print("built", p)
