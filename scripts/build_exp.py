"""Experimental library with -DOAI_EXP=<bits> (timing-only ablations of conv3_wino_sres's epilogue, unet_wino.h): python scripts/build_exp.py 1 2 3 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import build
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for bits in sys.argv[1:]:
    d = os.path.join(root, "build", "exp")
    os.makedirs(d, exist_ok=True)
    print(build.build_library(False, False, [f"-DOAI_EXP={bits}"], os.path.join(d, f"liboai_hip_exp{bits}.so"), os.path.join(d, f"_obj{bits}")))
