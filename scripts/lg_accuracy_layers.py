"""Where does the split path lose accuracy?  Token state after each half-layer vs an fp64 oracle."""
import importlib, sys, types, inspect
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
import oracle.lightglue_ref as RR
W = importlib.import_module("opencv-simpleslam_amd.weights")
LGm = importlib.import_module("opencv-simpleslam_amd.lightglue")
nat = importlib.import_module("opencv-simpleslam_amd._native")
N = 512
sd = W.random_lightglue_state_dict(5, match_gain=4.0, match_bias=3.0)
k0, d0, k1, d1 = lg_inputs.make_pair(N, seed=21)
torch.set_default_dtype(torch.float64)
src = inspect.getsource(RR).replace("torch.float32", "torch.float64")
mod = types.ModuleType("lg64"); exec(compile(src, "lg64", "exec"), mod.__dict__)
r64 = mod.lightglue_forward(sd, k0.astype(np.float64), d0.astype(np.float64), k1.astype(np.float64), d1.astype(np.float64),
                            {"depth_confidence": -1, "width_confidence": -1}, return_debug=True)
torch.set_default_dtype(torch.float32)
lg = LGm.LightGlueHIP(sd, max_kpts=N, depth_confidence=-1.0, width_confidence=-1.0)
Kc = lg.capacity
L = nat.lib()
for mode in ("f32", "f16x3"):
    lg.set_precision(mode)
    print(mode)
    for layer in (1, 2, 5, 9):
        for self_only in (1, 0):
            lg.debug_layers(layer, bool(self_only))
            lg.match(k0, d0, k1, d1, min_conf=0.0)
            x = lg.debug_read(0, (2, Kc, 256))
            key = "self" if self_only else "cross"
            ref = np.concatenate([r64["debug"]["layers"][layer - 1][key + "0"].numpy(), r64["debug"]["layers"][layer - 1][key + "1"].numpy()])
            xg = np.concatenate([x[0, :N], x[1, :N]])
            e = np.abs(xg - ref)
            print(f"   layer {layer} {key:5s}: max {e.max():.2e} rms {np.sqrt(np.mean(e**2)):.2e}")
