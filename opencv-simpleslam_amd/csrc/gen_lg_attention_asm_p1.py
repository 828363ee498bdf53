#!/usr/bin/env python3
"""The attention kernel of the opt-in precision "f16x3p1" (P as one fp16 plane in P.V; sslam_lightglue_set_precision(lg, 2)):
gen_lg_attention_asm.py with its P1 switch, as a code object of its own (lg_attention_asm_p1_kernel)."""
import os
import runpy
from pathlib import Path

os.environ["ATTN_ASM_P1"] = "1"
os.environ["ATTN_ASM_KERNEL"] = "lg_attention_asm_p1_kernel"
runpy.run_path(str(Path(__file__).with_name("gen_lg_attention_asm.py")), run_name="__main__")
