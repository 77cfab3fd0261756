"""Diagnostic: is the conv path power/clock-limited?  Times the 32-tile segmentation with real and all-zero data and samples
rocm-smi (sclk, power) while it loops."""
import os, sys, time, subprocess, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
prec = os.environ.get("PREC", "fp16x3")
for zero in (0, 1):
    sd = make_unet_state_dict(0)
    if zero:
        sd = {k: (v * 0 if ("weight" in k and v.ndim > 1) or "bias" in k else v) for k, v in sd.items()}
    eng = UNetEngine(sd, precision=prec)
    vol = torch.from_numpy(make_volume(0)).cuda() * (0 if zero else 1)
    samples = []
    stop = False
    def sampler():
        while not stop:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                s = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l]
                samples.append(" | ".join(x.split(":", 1)[-1].strip() for x in s))
            except Exception as e:
                samples.append(repr(e))
            time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    for rep in range(40):
        torch.cuda.synchronize(); t = time.time()
        eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, 32), 0, 32)
        torch.cuda.synchronize(); dt = time.time() - t
    stop = True; th.join()
    print(f"zero={zero} {prec} last rep {dt*1e3:.1f} ms")
    for s in samples[-4:]: print("   ", s)
