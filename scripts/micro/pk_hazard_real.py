"""The packed-fp32 victim of scripts/micro/pk_hazard.hip beside the REAL kernels of this library: full-size fp16x3 segmentations
(conv3_igemm_sres / sres2 of liboai_hip.so) on the main stream, the victim on a side stream.  Mismatch counts per operand form."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
so = "/tmp/pk_hazard.so"
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-DPK_SO", os.path.join(ROOT, "scripts/micro/pk_hazard.hip"), "-o", so], check=True)
lib = C.CDLL(so)
eng = UNetEngine(make_unet_state_dict(0), precision=os.environ.get("PREC", "fp16x3"))
vol = torch.from_numpy(make_volume(0)).cuda()
seg = lambda: eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), None, 0, 160, (8, 16, 16))
seg(); torch.cuda.synchronize()
side = torch.cuda.Stream()
counts = torch.zeros(16, dtype=torch.int32, device="cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for beside in (False, True):
    counts.zero_(); torch.cuda.synchronize()
    for r in range(rounds):
        if beside:
            seg()
        for k in range(16):
            lib.pk_victim(C.c_void_p(side.cuda_stream), 512, 1 << 12, C.c_void_p(counts.data_ptr()))
        torch.cuda.synchronize()
    c = counts.cpu().tolist()
    print(f"{'victim beside the segmentation (' + eng.precision + ')' if beside else 'victim alone'}: of {rounds * 16 * 512 * 256 * 4096:.3g} evaluations per form: plain fma {c[0]} | "
          f"fma op_sel:[0,1,0] {c[1]} (product term zero: {c[12]}) | fma op_sel:[1,0,0] {c[2]} | mul op_sel:[0,1] {c[3]} | by 16-lane group {c[8:12]} | scalar control {c[13]}", flush=True)
