"""Diagnostic build only (python -m oai_analysis_2_amd.build --diag; OAI_LIB_PATH=build/diag/liboai_hip_diag.so OAI_STAMPS=1):
per-wave cycle budget of conv3_igemm_sres's chunk loop, from s_memtime stamps at the phase boundaries, per layer."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import _lib
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
lib = C.CDLL(_lib.LIB_PATH)
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
for kv in os.environ.get("OPTIONS", "").split(","):          # e.g. OPTIONS=fuse_first=0; OAI_STAMP_LAYER=<layer index> (ec1 = 1) stamps one layer only
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
SET = int(os.environ.get("STAMP_SET", "0"))      # must match the -DOAI_STAMP_SET the diagnostic library was built with
names = ({0: "loop->bar1", 1: "barrier 1", 2: "DMA issue", 3: "DMA wait", 4: "barrier 2", 5: "27 taps", 6: "epilogue", 7: "prologue"} if SET == 0 else
         {0: "last tap -> 1st epilogue barrier", 1: "split + LDS image", 2: "barriers", 3: "copy-out stores", 4: "fused dc0 dots", 5: "fused pool stores",
          6: "head write + tail", 7: "prologue + whole chunk loop"})
for tiles in (32, 160):
    eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, tiles), 0, tiles)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    lib.oai_diag_stamps(out, 1)
    eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, tiles), 0, tiles)
    torch.cuda.synchronize()
    lib.oai_diag_stamps(out, 1)
    waves, chunks = out[8], out[9]
    tot = sum(out[i] for i in names)
    print(f"{tiles} tiles: {waves} waves, {chunks} wave-chunks, {tot / max(waves, 1):.0f} cycles per wave")
    for i, n in (names.items() if waves else ()):
        per_chunk = SET == 0 and i < 6
        per = out[i] / (chunks if per_chunk else waves)
        print(f"   {n:34s} {100 * out[i] / tot:5.1f} %   {per:9.0f} cycles per {'chunk' if per_chunk else 'block'}")
    if out[24]:
        un = ["prologue", "k loop (DMA ring + MFMAs)", "epilogue barriers", "split + LDS image", "copy-out stores"]
        ut = sum(out[16 + i] for i in range(5))
        print(f"   upconv2_igemm_sres: {out[24]} waves, {ut / out[24]:.0f} cycles per wave")
        for i, n in enumerate(un):
            print(f"      {n:30s} {100 * out[16 + i] / ut:5.1f} %   {out[16 + i] / out[24]:9.0f} cycles per block")
