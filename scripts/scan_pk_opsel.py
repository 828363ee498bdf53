"""Command line of opencv-simpleslam_amd/isa_guard.py: the static scan for the packed-fp32 form that returns a wrong low half beside a
wide-operand MFMA (profiles/r06_aggregate_rnorm_diagnosis.md, section 5).  usage: scan_pk_opsel.py [FILE ...]"""
import importlib.util, sys
from pathlib import Path
spec = importlib.util.spec_from_file_location("sslam_isa_guard", Path(__file__).resolve().parent.parent / "opencv-simpleslam_amd" / "isa_guard.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
sys.exit(m.main())
