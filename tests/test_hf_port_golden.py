"""Independent evidence for the LightGlue restatement: per-stage outputs of the HF `transformers`
port (a separately written implementation of the same published network), stored in
tests/golden/hf_lightglue.npz by tests/golden/make_hf_lightglue_golden.py.

  * CPU: oracle/lightglue_ref.py reproduces every stored stage - positional encoding, the self and
    cross blocks of layers 0 / 3 / 8 in full and (r04) EVERY half layer of the 9-layer forward on the first 16
    tokens of each image, token confidence, the log-assignment matrix, the match filter.
  * GPU (-m gpu): the HIP path reproduces the same stages through the C-ABI (token states after
    k layers via the debug hooks; final matches).

The HF port is not the dependency the reference pins, so this does not pin the oracle in the
sense of DESIGN.md section 2; it is the one check available here that no code of this repository
produced."""
import numpy as np
import pytest
import torch

from conftest import ROOT, load_pkg
from oracle import lightglue_ref as L

G = np.load(ROOT / "tests" / "golden" / "hf_lightglue.npz")
NOCTL = {"depth_confidence": -1.0, "width_confidence": -1.0}       # all layers, no pruning (as the fixture)
SD_KW = dict(match_gain=4.0, match_bias=3.0)


@pytest.fixture(scope="module")
def sd():
    W = load_pkg("weights")
    s = W.random_lightglue_state_dict(int(G["seed"]), **SD_KW)
    chk = sum(float(np.abs(np.asarray(v, np.float64)).sum()) for v in s.values())
    assert abs(chk - float(G["weight_checksum"])) <= 1e-9 * chk, "seeded weights differ from the fixture's"
    return s


def test_oracle_reproduces_every_hf_stage(sd):
    out = L.lightglue_forward(sd, G["k0"], G["d0"], G["k1"], G["d1"], NOCTL, return_debug=True)
    d = out["debug"]
    np.testing.assert_allclose(d["kn0"].numpy(), G["kn"][0], atol=1e-6)
    np.testing.assert_allclose(d["x_in0"].numpy(), G["x_in"][0], atol=1e-5)
    # rotary tables: HF (cos, sin) [2, N, 64] == posenc()[:, 0, 0]
    enc = L.posenc({k: torch.as_tensor(v) for k, v in sd.items()}, torch.as_tensor(G["kn"]))
    np.testing.assert_allclose(enc[0, :, 0].numpy(), G["cos"], atol=1e-6)
    np.testing.assert_allclose(enc[1, :, 0].numpy(), G["sin"], atol=1e-6)
    for key in G.files:
        if key.startswith("self_") or key.startswith("cross_"):
            kind, i = key.split("_")
            for img in (0, 1):
                np.testing.assert_allclose(d["layers"][int(i)][f"{kind}{img}"].numpy(), G[key][img], atol=2e-5, rtol=1e-5,
                                           err_msg=key)
        if key.startswith("rows_"):                    # r04: every half layer of the full forward (first 16 tokens per image)
            _, kind, i = key.split("_")
            for img in (0, 1):
                np.testing.assert_allclose(d["layers"][int(i)][f"{kind}{img}"].numpy()[:G[key].shape[1]], G[key][img], atol=2e-5,
                                           rtol=1e-5, err_msg=key)
        if key.startswith("conf_"):
            i = int(key.split("_")[1])
            x = torch.stack([d["layers"][i]["cross0"], d["layers"][i]["cross1"]])
            np.testing.assert_allclose(L.token_confidence({k: torch.as_tensor(v) for k, v in sd.items()}, i, x).numpy(),
                                       G[key], atol=1e-5)
    assert out["stop"] == 9
    np.testing.assert_allclose(d["log_scores"].numpy(), G["log_scores"], atol=1e-4, rtol=2e-5)   # |log score| ~ 100
    # match filter: HF's matches0 (-1 = unmatched) vs the oracle's [K,2] list
    m0 = G["matches0"]
    want = np.stack([np.flatnonzero(m0 > -1), m0[m0 > -1]], 1)
    np.testing.assert_array_equal(out["matches"].numpy(), want)
    np.testing.assert_allclose(out["scores"].numpy(), G["mscores0"][m0 > -1], atol=1e-4)     # exp of a log score accurate to ~1e-4
    assert len(want) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x3p1", "f16x3", "f32"])       # (f16x3p1: the shipped default)
def test_hip_path_reproduces_the_hf_stages(sd, gpu_ctx, precision):
    LG = load_pkg("lightglue").LightGlueHIP
    n = int(G["n"])
    lg = LG(sd, max_kpts=n, ctx=gpu_ctx, depth_confidence=-1.0, width_confidence=-1.0)
    lg.set_precision(precision)
    Kc = lg.capacity
    args = (G["k0"], G["d0"], G["k1"], G["d1"])
    # token states: fp32-grade (3e-5) in exact fp32 and in "f16x3"; the shipped default carries P into P.V as ONE fp16 plane and
    # is 4.6e-5 from the HF port at the last layer (r06, this fixture) - inside north_star's 1e-3 by a factor of 20, held to 1e-4
    atol = 1e-4 if precision == "f16x3p1" else 3e-5
    for key in sorted(G.files):
        if not (key.startswith("self_") or key.startswith("cross_") or key.startswith("rows_")):
            continue
        kind, i = key.split("_")[-2:]
        lg.debug_layers(int(i) + 1, self_only=(kind == "self"))
        lg.match(*args, min_conf=0.0)
        x = lg.debug_read(0, (2, Kc, 256))
        rows = G[key].shape[1]                            # whole images for the stored stages, 16 tokens for every half layer
        for img in (0, 1):
            np.testing.assert_allclose(x[img, :rows], G[key][img], atol=atol, rtol=1e-5, err_msg=f"{key} {precision}")
    lg.debug_layers(9, False)
    ij, sc, stop = lg.match(*args, min_conf=0.0)
    m0 = G["matches0"]
    want = np.stack([np.flatnonzero(m0 > -1), m0[m0 > -1]], 1)
    np.testing.assert_array_equal(ij, want)                       # index arrays identical to the HF port's
    np.testing.assert_allclose(sc, G["mscores0"][m0 > -1], atol=1e-4)
    assert stop == 9
    lg.close()
