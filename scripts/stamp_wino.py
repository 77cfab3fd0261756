"""Diagnostic build only (python -m oai_analysis_2_amd.build --diag; OAI_LIB_PATH=build/diag/liboai_hip_diag.so OAI_STAMPS=1
OAI_STAMP_LAYER=<k>): where a workgroup of conv3_wino_sres spends its time on layer k (ec3 = 3 .. ec7 = 7, dc8 = 9, dc7 = 10, dc5 = 12,
dc4 = 13, dc2 = 15) -- prologue / chunk loop / epilogue per wave from s_memrealtime stamps (100 MHz), and what the CUs spend BETWEEN
workgroups: span of the launch x 256 CUs - the workgroups' own time (meaningful for the layers that are one launch)."""
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
waves = out[8]
if not waves:
    print("no stamps (layer not in the Winograd form?)"); sys.exit(0)
pro, loop, epi = (out[i] / waves / 100.0 for i in range(3))                     # us per wave
exch, img, cpy = (out[i] / waves / 100.0 for i in (3, 4, 5))
start = (~out[10]) & 0xFFFFFFFFFFFFFFFF
span = (out[11] - start) / 100.0
wpb = int(os.environ.get("WAVES_PER_WG", "8"))                                 # stamping waves per workgroup: 8 two-group, 4 multipliers of the specialised form
wgs = waves / wpb
own = (pro + loop + epi) * wgs / 256.0                                        # us of workgroup time per CU
print(f"layer {os.environ.get('OAI_STAMP_LAYER')}: {waves} waves = {wgs:.0f} workgroups ({wgs / 256:.1f} per CU); per workgroup: prologue {pro:.1f} us, "
      f"chunk loop {loop:.1f} us, epilogue {epi:.1f} us (exchange {exch:.1f}, output transform + image {img:.1f}, copy-out + pool {cpy:.1f}) = {pro + loop + epi:.1f} us; span of the launch(es) {span:.0f} us, workgroup time per CU {own:.0f} us "
      f"-> between workgroups {span - own:.0f} us = {(span - own) / max(wgs / 256, 1):.1f} us per workgroup")
