"""The C-ABI from plain C: the header is valid C99 (CPU), and a gcc-built client without Python,
torch or C++ drives three entry points on the GPU (tests/c_abi_client.c)."""
import os
import subprocess

import pytest

from conftest import ROOT

LIBDIR = ROOT / "opencv-simpleslam_amd" / "lib"


def test_header_is_plain_c99():
    out = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-Werror",
                          str(ROOT / "include" / "sslam_hip.h")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_c_client_links_against_the_library(tmp_path):
    """Link only (no GPU needed): every symbol the client uses resolves against libsslam_hip.so."""
    if not (LIBDIR / "libsslam_hip.so").exists():
        pytest.skip("library not built")
    exe = tmp_path / "c_client"
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_abi_client.c"),
                          "-L", str(LIBDIR), "-lsslam_hip", "-lm", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu(tmp_path):
    exe = tmp_path / "c_client"
    subprocess.run(["gcc", "-std=c99", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_abi_client.c"),
                    "-L", str(LIBDIR), "-lsslam_hip", "-lm", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, LD_LIBRARY_PATH=f"{LIBDIR}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "c client ok" in out.stdout
