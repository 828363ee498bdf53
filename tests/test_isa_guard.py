"""The build-time guard against the one instruction form known to compute silently wrong values on gfx950 beside other queues' MFMA
kernels (opencv-simpleslam_amd/isa_guard.py; profiles/r06_aggregate_rnorm_diagnosis.md section 5): the scanner recognises the form,
the built product contains none, and a library that does is refused."""
import importlib
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("sslam_isa_guard", ROOT / "opencv-simpleslam_amd" / "isa_guard.py")
guard = importlib.util.module_from_spec(spec)
spec.loader.exec_module(guard)


@pytest.mark.parametrize("text, kind", [
    ("v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]", "bad"),          # the instruction of al_aggregate_kernel
    ("v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[0,1]", "bad"),
    ("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,1] op_sel_hi:[1,0,1]", "bad"),
    ("v_pk_add_f32 v[2:3], s[4:5], v[6:7] op_sel:[0,1] // 000000001A2C: D3B24002 1802", "bad"),
    ("v_pk_add_f32 v[10:11], v[8:9], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]", "same"),               # horizontal add: never failed
    ("v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[1,0] op_sel_hi:[0,1]", "pk"),             # the mirrored form: never failed
    ("v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel_hi:[1,0]", "pk"),
    ("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,0,1]", "pk"),
    ("v_pk_mul_f16 v1, v2, v3 op_sel:[0,1]", ""),
    ("v_mul_f32_e32 v1, v2, v3", ""),
])
def test_classify(text, kind):
    assert guard.classify(text) == kind


LISTING = """
_ZN1a6kernelEv:
\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5]
\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]
.LBB0_1:
\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]
\ts_endpgm
.Lfunc_end0:
_ZN1a5otherEv:
\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]
\tv_mfma_f32_32x32x8_f16 v[0:15], v[4:5], v[8:9], v[0:15]
"""


def test_scan_text_and_check(tmp_path):
    res = guard.scan_text(LISTING.split("\n"))
    assert res["_ZN1a6kernelEv"] == dict(pk=2, bad=1, same=0, wide=1, bad_text=["v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]"])
    assert res["_ZN1a5otherEv"]["bad"] == 0 and res["_ZN1a5otherEv"]["same"] == 1 and res["_ZN1a5otherEv"]["wide"] == 0
    f = tmp_path / "x.s"
    f.write_text(LISTING)
    with pytest.raises(RuntimeError, match="op_sel:\\[0,1\\]"):
        guard.check([f])
    g = tmp_path / "y.s"
    g.write_text(LISTING.replace("op_sel:[0,1,0]", "op_sel:[1,0,0]"))
    guard.check([g])


def test_product_library_holds_no_such_instruction():
    importlib.import_module("opencv-simpleslam_amd.build").build_native()
    files = guard.product_files()
    assert files[0].name == "libsslam_hip.so" and files[0].exists()
    res = guard.scan(files)
    kernels = res["libsslam_hip.so"]
    assert sum(v["pk"] for v in kernels.values()) > 1000          # the scan SAW the library's packed instructions ...
    assert sum(v["wide"] for v in kernels.values()) > 1000        # ... and its wide-operand MFMAs (it runs beside its own triggers)
    assert guard.failing(res) == []
    agg = [v for k, v in kernels.items() if "al_aggregate_kernel" in k]
    assert agg == [] or all(v["pk"] == 0 for v in agg)            # the kernel the fault was found in: no packed fp32 at all
