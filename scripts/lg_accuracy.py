"""Accuracy of the two LightGlue arithmetic paths against the torch-CPU oracle (fp32) and against an
fp64 evaluation of the oracle."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
from oracle import lightglue_ref as R
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sd = W.random_lightglue_state_dict(5, match_gain=4.0, match_bias=3.0)
k0, d0, k1, d1 = lg_inputs.make_pair(N, seed=21)
ref = R.lightglue_forward(sd, k0, d0, k1, d1, return_debug=True)
x_ref = torch.cat([ref["debug"]["x_out0"], ref["debug"]["x_out1"]]).numpy()
# fp64 oracle: same code with float64 tensors
sd64 = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in sd.items()}
orig = torch.float32
import oracle.lightglue_ref as RR
def fwd64():
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    old = torch.get_default_dtype(); torch.set_default_dtype(torch.float64)
    try:
        # monkeypatch: as_tensor(dtype=float32) inside lightglue_forward -> use a float64 clone of the function
        import types, inspect
        src = inspect.getsource(RR).replace("torch.float32", "torch.float64")
        mod = types.ModuleType("lg64"); exec(compile(src, "lg64", "exec"), mod.__dict__)
        return mod.lightglue_forward(sd, k0.astype(np.float64), d0.astype(np.float64), k1.astype(np.float64), d1.astype(np.float64), return_debug=True)
    finally:
        torch.set_default_dtype(old)
r64 = fwd64()
x64 = torch.cat([r64["debug"]["x_out0"], r64["debug"]["x_out1"]]).numpy()
print(f"N={N}: torch-CPU fp32 oracle vs fp64: max|dx| = {np.abs(x_ref - x64).max():.3e}  rms = {np.sqrt(np.mean((x_ref-x64)**2)):.3e}  (|x| rms {np.sqrt(np.mean(x64**2)):.3f})")
lg = LG(sd, max_kpts=N)
Kc = lg.capacity
for mode in ("f32", "f16x3"):
    lg.set_precision(mode)
    ij, sc, stop = lg.match(k0, d0, k1, d1, min_conf=0.0)
    x = lg.debug_read(0, (2, Kc, 256))
    xg = np.concatenate([x[0, :N], x[1, :N]])
    sim = lg.debug_read(1, (Kc, Kc))[:N, :N]
    print(f"  HIP {mode:6s}: vs fp64 max|dx| = {np.abs(xg - x64).max():.3e} rms = {np.sqrt(np.mean((xg-x64)**2)):.3e} | vs torch fp32 max|dx| = {np.abs(xg - x_ref).max():.3e} | sim max|d| vs fp64 = {np.abs(sim - r64['debug']['sim'].numpy()).max():.3e} | matches equal oracle: {np.array_equal(ij, ref['matches'].numpy())}")
