import os, sys, subprocess
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
b = eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, 160), 0, 32, (8, 16, 16))
m = eng.stitch(b, vol.shape, (32, 128, 128), (8, 16, 16), (8, 16, 16))
np.save(sys.argv[1], m.cpu().numpy())
