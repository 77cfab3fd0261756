"""Quick device timing of the segmentation path (not the bench contract; see bench.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine

ntiles = int(os.environ.get("NTILES", "160"))
batch = int(os.environ.get("BATCH", "16"))
eng = UNetEngine(make_unet_state_dict(0))
vol = torch.from_numpy(make_volume(0)).cuda()
tile, ovl = (32, 128, 128), (8, 16, 16)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    b = eng.segment_tiles(vol, tile, ovl, (0, ntiles), 0, batch, (8, 16, 16) if int(os.environ.get('CROP', '1')) else None)
    torch.cuda.synchronize(); dt = time.time() - t
    fl_t = eng.tile_flops(tile, ovl, True) * ntiles; fl_f = eng.tile_flops(tile, ovl, False) * ntiles
    notrim = bool(int(os.environ.get("OAI_NO_TRIM", "0")))
    print(f"variant={os.environ.get('OAI_CONV_VARIANT','0')} notrim={notrim} tiles={ntiles} batch={batch} time={dt:.3f}s "
          f"algorithmic {fl_t/dt/1e12:.1f} TF/s, executed-untrimmed-equivalent {fl_f/dt/1e12:.1f} TF/s, vol/s={ntiles/160/dt:.3f}")
