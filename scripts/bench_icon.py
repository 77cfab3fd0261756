"""Device time of one ICON direction at 80x192x192 (what ICON_Registration.register runs per volume)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_volume
eng = IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192))
A = torch.from_numpy(make_volume(1, (80, 192, 192))).cuda(); B = torch.from_numpy(make_volume(2, (80, 192, 192))).cuda()
for _ in range(3): eng.phi(A, B)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): phi = eng.phi(A, B)
e1.record(); torch.cuda.synchronize()
print(f"ICON one direction 80x192x192: {e0.elapsed_time(e1) / 10:.2f} ms, checksum {float(phi.double().sum()):.6f}")
