"""al_aggregate_kernel ALONE as the victim (scripts/ubench/agg_victim.hip): N streams loop the kernel on fixed inputs, every launch
compared on the device with the first one, beside an aggressor of scripts/aggressor_util.py.
usage: agg_victim_run.py LIB[:CODE_OBJECT] [beside=lightglue:ring,noasm] [rounds=300] [iters=40] [streams=1] [F=2] [check_every=1]
LIB:CODE_OBJECT launches the kernel of a patched code object (scripts/agg_isa_patch.py) instead of the compiled-in one.
Prints the number of rnorm / s8 words that ever differed and the first events (index -> frame row, column, lane)."""
import ctypes, importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
LIB = sys.argv[1]
BESIDE = sys.argv[2] if len(sys.argv) > 2 else "lightglue:ring,noasm"
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 40
NS = int(sys.argv[5]) if len(sys.argv) > 5 else 1
F = int(sys.argv[6]) if len(sys.argv) > 6 else 2
EVERY = int(sys.argv[7]) if len(sys.argv) > 7 else 1
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
nat = pkg._native
LIB, _, CO = LIB.partition(":")
V = ctypes.CDLL(LIB)
KERNEL = b"_ZN12_GLOBAL__N_119al_aggregate_kernelENS_3PyrEPKfPfS3_m"
V.victim_create.restype = ctypes.c_void_p
V.victim_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
V.victim_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
V.victim_poll.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
V.victim_destroy.argtypes = [ctypes.c_void_p]
HP, WP = 384, 1248
from aggressor_util import make_aggressor
aggr_ctx, aggressor = make_aggressor(BESIDE, nat, W, ROOT)
ctxs = [nat.Context(0) for _ in range(NS)]
vic = [V.victim_create(HP, WP, F, 7 + j) for j in range(NS)]
assert all(vic), "victim_create failed"
import os
REF_PATCHED = os.environ.get("VICTIM_REF_PATCHED") == "1"       # probe variants store another value than 1 / ||F||: their own first launch is the reference
def use_module():
    V.victim_use_module.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    rc = V.victim_use_module(CO.encode(), KERNEL)
    assert rc == 0, f"victim_use_module({CO}) = {rc}"
if CO and REF_PATCHED:
    use_module()
for j in range(NS):                                   # the reference outputs: one launch each with nothing else on the GPU
    aggr_ctx.sync()
    assert V.victim_run(vic[j], ctypes.c_void_p(int(ctxs[j].stream)), 1, 0) == 0
    ctxs[j].sync()
if CO and not REF_PATCHED:                            # (the reference is the COMPILED-IN kernel's: a patch that changes a value shows as every word differing)
    use_module()
t0 = time.time()
for r in range(ROUNDS):
    aggressor()
    for j in range(NS):
        assert V.victim_run(vic[j], ctypes.c_void_p(int(ctxs[j].stream)), ITERS, EVERY) == 0
    if r % 50 == 49:
        for c in ctxs: c.sync()
        aggr_ctx.sync()
        print(f"  round {r + 1}: {time.time() - t0:.1f} s", flush=True)
tot_rn = tot_s8 = 0
for j in range(NS):
    out = np.zeros(64, np.uint32)
    assert V.victim_poll(vic[j], ctypes.c_void_p(int(ctxs[j].stream)), out.ctypes.data) == 0
    tot_rn += int(out[0]); tot_s8 += int(out[1])
    for k in range(min(int(out[0]), 15)):
        i, got, ref, launch = (int(x) for x in out[4 + 4 * k: 8 + 4 * k])
        if "pkprobe" in LIB:                             # scripts/ubench/pk_probe.hip: lane | half << 8 | workgroup << 16
            print(f"  stream {j}: launch {launch} workgroup {i >> 16} lane {i & 63} (16-lane group {(i & 63) // 16}) {'high' if (i >> 8) & 1 else 'low'} half: "
                  f"got {float(np.uint32(got).view(np.float32))!r} (0x{got:08x}) expected {float(np.uint32(ref).view(np.float32))!r}", flush=True)
            continue
        if "insnprobe" in LIB:                           # scripts/ubench/insn_probe.hip: the thread whose hash differs
            print(f"  stream {j}: launch {launch} thread {i} lane {i % 64} (16-lane group {(i % 64) // 16}): hash 0x{got:08x}, alone 0x{ref:08x}", flush=True)
            continue
        y, x = divmod(i, WP)
        g, rf = np.uint32(got).view(np.float32), np.uint32(ref).view(np.float32)
        print(f"  stream {j}: launch {launch} row {y} col {x} (lane {x % 64}, 16-lane group {(x % 64) // 16}) got {float(g)!r} (0x{got:08x}) ref {float(rf)!r} "
              + (f"n2 ratio {float(rf) ** 2 / float(g) ** 2:.5f}" if g != 0 and not REF_PATCHED else ""), flush=True)
aggr_ctx.sync()
print(f"{(CO or LIB).split('/')[-1]} beside {BESIDE}: {ROUNDS} rounds x {ITERS} launches x {NS} streams x {F} frames = {ROUNDS * ITERS * NS} launches "
      f"in {time.time() - t0:.1f} s: rnorm words differing {tot_rn} (runs of 16: {tot_rn / 16:.1f}), s8 words differing {tot_s8}", flush=True)
for h in vic: V.victim_destroy(h)
