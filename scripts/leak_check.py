import sys, gc
sys.path.insert(0, "/root/repo")
import torch
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.synth import make_unet_state_dict, make_icon_state_dict, make_volume
sd, isd = make_unet_state_dict(0), make_icon_state_dict(0, 0.1)
vol = torch.from_numpy(make_volume(0, (40, 100, 100))).cuda()
free0 = None
for i in range(12):
    e = UNetEngine(sd, precision="fp16x3" if i % 2 else "f32")
    b = e.segment_tiles(vol, (32, 128, 128), (8, 16, 16), None, 0, 4, (8, 16, 16))
    ic = IconEngine(isd, (40, 48, 48))
    del e, b, ic
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if i == 1: free0 = free
    if i in (1, 11): print(i, "free GiB", free / 2**30)
print("leak per cycle (MiB):", (free0 - free) / 10 / 2**20)
