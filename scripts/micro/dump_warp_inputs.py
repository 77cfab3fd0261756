"""Writes the 160^3 warp bench inputs of scripts/bench_warp.py as raw float32 for scripts/micro/warp_probe.hip."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oai_analysis_2_amd.synth import identity_map, make_smooth_field, make_volume
shape = (160, 160, 160)
make_volume(1, shape).astype(np.float32).tofile("/tmp/warp_src.bin")
(torch.from_numpy(identity_map(shape)) + torch.from_numpy(make_smooth_field(4, shape, 0.03))).numpy().astype(np.float32).tofile("/tmp/warp_coords.bin")
