"""Diagnostic build only (python -m oai_analysis_2_amd.build --diag; OAI_LIB_PATH=build/diag/liboai_hip_diag.so OAI_STAMPS=1 OAI_STAMP_LAYER=15
OPTIONS=persistent=1): where the persistent form of conv3_wino_sres (dc2) spends a block -- multiplying wave 0: block switch, chunk loop (and the part of
it spent at the chunk-end barriers = waiting for the staging waves), epilogue; staging wave 4: its waits at the chunk-end barriers (slack) and its epilogue."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import _lib
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
lib = C.CDLL(_lib.LIB_PATH)
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
for kv in os.environ.get("OPTIONS", "").split(","):
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
tiles = 160
for _ in range(2):
    eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, tiles), 0, tiles)
    torch.cuda.synchronize()
out = (C.c_ulonglong * 32)()
lib.oai_diag_stamps(out, 1)
eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, tiles), 0, tiles)
torch.cuda.synchronize()
lib.oai_diag_stamps(out, 1)
blocks = out[8]
if not blocks:
    print("no stamps"); sys.exit(0)
us = lambda i: out[i] / blocks / 100.0
print(f"{blocks} blocks; per block, multiplying wave 0: switch {us(0):.2f} us, chunk loop {us(1) + us(3):.2f} us of which {us(3):.2f} at the chunk-end barriers, epilogue {us(2):.2f} us "
      f"= {us(0) + us(1) + us(2) + us(3):.2f} us; staging wave 4: at the chunk-end barriers {us(4):.2f} us, staging work {us(6):.2f} us, epilogue {us(5):.2f} us")
