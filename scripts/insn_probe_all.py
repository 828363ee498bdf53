"""Every form of scripts/ubench/insn_probe.hip in ONE process: per form, a solitary reference launch, then `rounds` x `iters` launches
beside the aggressor, each compared on the device with the reference.  Prints the forms that differ and a summary.
usage: insn_probe_all.py [beside=synthetic:mfma_16x16x32_f16] [rounds=40] [iters=10] [first=0] [last=-1]"""
import ctypes, importlib, subprocess, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
BESIDE = sys.argv[1] if len(sys.argv) > 1 else "synthetic:mfma_16x16x32_f16"
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
FIRST = int(sys.argv[4]) if len(sys.argv) > 4 else 0
LAST = int(sys.argv[5]) if len(sys.argv) > 5 else -1
so = ROOT / "scripts" / "ubench" / "libinsnprobe.so"
if not so.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", str(so),
                    str(ROOT / "scripts" / "ubench" / "insn_probe.hip")], check=True, capture_output=True)
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
nat = pkg._native
V = ctypes.CDLL(str(so))
V.victim_create.restype = ctypes.c_void_p
V.victim_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
V.victim_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
V.victim_poll.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
V.victim_destroy.argtypes = [ctypes.c_void_p]
V.victim_mode_text.restype = ctypes.c_char_p
nprod = ctypes.c_int(0)
NM = V.victim_modes(ctypes.byref(nprod))
from aggressor_util import make_aggressor
aggr_ctx, aggressor = make_aggressor(BESIDE, nat, W, ROOT)
ctx = nat.Context(0)
stream = ctypes.c_void_p(int(ctx.stream))
last = NM - 1 if LAST < 0 else LAST
t0 = time.time()
differing, clean, threads = [], 0, 0
for m in range(FIRST, last + 1):
    h = V.victim_create(0, 0, m, 0)
    assert h, m
    aggr_ctx.sync()
    assert V.victim_run(h, stream, 1, 0) == 0          # the reference: alone
    ctx.sync()
    for r in range(ROUNDS):
        aggressor()
        assert V.victim_run(h, stream, ITERS, 0) == 0
    out = np.zeros(64, np.uint32)
    assert V.victim_poll(h, stream, out.ctypes.data) == 0
    aggr_ctx.sync()
    V.victim_destroy(h)
    threads += ROUNDS * ITERS * 2048 * 256
    text = V.victim_mode_text(m).decode().replace("\n\t", " ; ")
    if out[0]:
        differing.append((m, text, int(out[0])))
        lanes = sorted({int(out[4 + 4 * k]) % 64 // 16 for k in range(min(int(out[0]), 15))})
        print(f"  DIFFERS  {m:4d}{' (extra)' if m >= nprod.value else ''}  {text}: {int(out[0])} thread hashes, 16-lane groups of the first ones {lanes}", flush=True)
    else:
        clean += 1
    if (m - FIRST) % 50 == 49:
        print(f"  ... form {m}, {time.time() - t0:.0f} s", flush=True)
print(f"beside {BESIDE}: forms {FIRST}..{last} ({nprod.value} of the product, {NM - nprod.value} extra): {clean} never differed in "
      f"{ROUNDS * ITERS} launches x 524 288 threads x 256 instructions each; {len(differing)} differed: {[d[0] for d in differing]}", flush=True)
