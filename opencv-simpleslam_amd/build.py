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


def source_digest() -> str:
    """12 hex digits over everything the library is built from (csrc/*.hip, *.hpp, the assembly generators, the C-ABI header),
    names and contents: what a profile or a bench line records to say WHICH kernels it measured (the snapshot on the GPU box
    has no .git), and what tests/test_profiles_fresh.py compares with the digest stored beside the committed profiles."""
    files = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("gen_*.py")) + [PKG_DIR.parent / "include" / "sslam_hip.h"]
    return _digest(*[x for f in files for x in (f.name, f.read_bytes())])[:12]


def _llvm_bin() -> Path:
    """clang / ld.lld of the SAME ROCm install as the hipcc in use (ROCM_PATH, else next to the resolved hipcc)."""
    roots = [os.environ.get("ROCM_PATH"), str(Path(os.path.realpath(_hipcc())).parent.parent), "/opt/rocm"]
    for r in filter(None, roots):
        for sub in ("lib/llvm/bin", "llvm/bin"):
            d = Path(r) / sub
            if (d / "clang").exists() and (d / "ld.lld").exists():
                return d
    raise RuntimeError("clang / ld.lld of the ROCm install not found (set ROCM_PATH)")



def _build_asm_kernels(obj_dir: Path, force: bool, verbose: bool):
    """Hand-written gfx950 assembly kernels: csrc/gen_<name>.py prints the assembly text (the generator IS the
    source: register map, schedule, hazards are written there), it is assembled and linked into a code object
    (clang -x assembler, ld.lld -shared) and the code object is embedded into the host library as the byte array
    `sslam_<name>_hsaco` (an .incbin stub), which the library loads with hipModuleLoadData at instance creation."""
    llvm = _llvm_bin()
    clang, lld = llvm / "clang", llvm / "ld.lld"
    out = []
    for gen in sorted(CSRC.glob("gen_*.py")):
        name = gen.stem[4:]
        obj = obj_dir / f"{name}_hsaco.o"
        keyf = obj_dir / f"{name}_hsaco.key"
        key = _digest(clang, lld, ARCH, *[g_.read_bytes() for g_ in sorted(CSRC.glob("gen_*.py"))])     # (a generator may run another)
        if force or not obj.exists() or not keyf.exists() or keyf.read_text().strip() != key:
            keyf.unlink(missing_ok=True)
            env = {k: v for k, v in os.environ.items() if not k.startswith("ATTN_ASM_")}     # no experiment switches
            text = subprocess.run([sys.executable, str(gen)], capture_output=True, text=True, env=env)
            if text.returncode != 0:
                raise RuntimeError(f"{gen.name} failed:\n{text.stderr}")
            asm, dev_o, hsaco, stub = (obj_dir / f"{name}{e}" for e in (".s", ".dev.o", ".hsaco", "_stub.s"))
            asm.write_text(text.stdout)
            stub.write_text(f'    .section .rodata\n    .globl sslam_{name}_hsaco\n    .p2align 12\nsslam_{name}_hsaco:\n'
                            f'    .incbin "{hsaco}"\n    .globl sslam_{name}_hsaco_end\nsslam_{name}_hsaco_end:\n'
                            f'    .section .note.GNU-stack,"",@progbits\n')
            for cmd in ([str(clang), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", str(asm), "-o", str(dev_o)],
                        [str(lld), "-shared", str(dev_o), "-o", str(hsaco)],
                        [str(clang), "-c", "-fPIC", str(stub), "-o", str(obj)]):
                if verbose:
                    print(" ".join(cmd), flush=True)
                res = subprocess.run(cmd, capture_output=True, text=True)
                if res.returncode != 0:
                    raise RuntimeError(f"assembling {name} failed:\n{res.stdout}\n{res.stderr}")
            keyf.write_text(key + "\n")
        out.append((obj, key))
    return out


def _isa_guard():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sslam_isa_guard", PKG_DIR / "isa_guard.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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
    if extra and os.environ.get("SSLAM_EXPERIMENT_BUILD") != "1":
        # the ablation switches in csrc/ (FFN_ABL, SSLAM_DBG_NOMFMA, AL_B2_ASM, AL_AGG_FAST_SELU ...) change ARITHMETIC: a flag
        # variable left in the environment must not leak into a product build.  The A/B scripts under scripts/ say so.
        raise RuntimeError(f"SSLAM_EXTRA_HIPCC_FLAGS={' '.join(extra)!r} is set but SSLAM_EXPERIMENT_BUILD is not 1: refusing to build "
                           "the product library with experiment flags (unset the variable, or export SSLAM_EXPERIMENT_BUILD=1 "
                           "for an A/B run under scripts/)")
    # (the compiler's own version string is part of every key: another hipcc at the same path rebuilds everything - the code
    #  shapes the stress tests vouch for are those of the compiler that built the library under test)
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    base_key = _digest(hipcc, ver, *HIPCC_FLAGS, "|", *extra, *[h.read_bytes() for h in headers])

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
    built += _build_asm_kernels(obj_dir, force, verbose)
    objs = [o for o, _ in built]

    lib_keyf = LIB_DIR / "libsslam_hip.key"
    lib_key = _digest(*[k for _, k in built])
    if force or not LIB_PATH.exists() or not lib_keyf.exists() or lib_keyf.read_text().strip() != lib_key:
        lib_keyf.unlink(missing_ok=True)
        # -z defs: an undefined symbol is a link error here, not a dlopen failure on the GPU box (hipcc's host pass can
        # drop the definition of a kernel template specialization without a diagnostic)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-z,defs", "-o", str(LIB_PATH),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
        # r06: no product library with a packed-fp32 instruction of the one form that silently computes a wrong value beside
        # another queue's MFMA kernels (isa_guard.py; the compiler forms it whenever its cost model likes).  Experiment builds
        # (the scripts that REPRODUCE the fault) are exempt.
        if os.environ.get("SSLAM_EXPERIMENT_BUILD") != "1" and os.environ.get("SSLAM_SKIP_ISA_GUARD") != "1":
            try:
                _isa_guard().check([LIB_PATH, *sorted(obj_dir.glob("*.hsaco"))])
            except RuntimeError:
                LIB_PATH.unlink(missing_ok=True)
                raise
            except (OSError, subprocess.CalledProcessError) as e:      # the disassembler itself is missing or failed: fail closed, say how to opt out
                LIB_PATH.unlink(missing_ok=True)
                raise RuntimeError(f"isa_guard could not disassemble the library ({e!r}): it needs llvm-objdump of the ROCm install under "
                                   "/opt/rocm/lib/llvm/bin; SSLAM_SKIP_ISA_GUARD=1 builds without the check (not for a product build)") from e
        lib_keyf.write_text(lib_key + "\n")
    return LIB_PATH


if __name__ == "__main__":
    if "--digest" in sys.argv:
        print(source_digest())
        sys.exit(0)
    p = build_native(force="--force" in sys.argv, verbose=True)
    print("built", p)
