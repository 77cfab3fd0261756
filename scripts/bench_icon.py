"""Device time of one ICON direction at 80x192x192 (what ICON_Registration.register runs per volume), per step tree
(oai_analysis_2_amd.synth.ICON_TREES): `python scripts/bench_icon.py [3step 4step multires multires4]`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_volume
A = torch.from_numpy(make_volume(1, (80, 192, 192))).cuda(); B = torch.from_numpy(make_volume(2, (80, 192, 192))).cuda()
for tree in (sys.argv[1:] or ["3step", "4step", "multires", "multires4"]):
    eng = IconEngine(make_icon_state_dict(0, 0.05, tree), (80, 192, 192))
    for _ in range(3): eng.phi(A, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): phi = eng.phi(A, B)
    e1.record(); torch.cuda.synchronize()
    print(f"ICON one direction 80x192x192, {tree} = {eng.tree.describe()} (U-Nets per level {eng.describe()[1][:3]}): "
          f"{e0.elapsed_time(e1) / 10:.2f} ms, checksum {float(phi.double().sum()):.6f}")
