"""Diagnostic 5: alternate two volumes through the overlapped pipeline; which workspace intermediate first deviates from the clean run?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

shape = (160, 384, 384)
D, H, W = 80, 192, 192
vh, vl = D * H * W, (D // 2) * (H // 2) * (W // 2)
al = lambda n: (n * 4 + 255) // 256 * 256
names, sizes = ["a", "b", "d1", "d2", "aw", "d3", "Aw"], [vl, vl, 3 * vl, 3 * vl, vl, 3 * vh, vh]
atlas = Image(make_volume(1000, shape), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
vols = [torch.from_numpy(make_volume(i, shape)).cuda() for i in range(2)]
meta = Image(make_volume(0, shape), [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])
icon = IconEngine(make_icon_state_dict(0, last_scale=0.1))
icon.set_graph(False)

def snapshot():
    out, o = {}, 0
    for n, s in zip(names, sizes):
        out[n] = icon._ws[o:o + 4 * s].view(torch.float32).clone()
        o += al(s)
    return out

unet = UNetEngine(make_unet_state_dict(0), precision=os.environ.get("PREC", "fp16x3"))
pipe = VolumePipeline(unet, icon, atlas)
clean = []
for v in vols:
    p = pipe.register(v); torch.cuda.synchronize(); clean.append((p.clone(), snapshot()))
for trial in range(12):
    k = trial % 2
    r = pipe.run(vols[k], meta); torch.cuda.synchronize(); dirty = snapshot()
    bad = {n: (dirty[n] - clean[k][1][n]).abs() for n in names}
    dphi = (r.phi - clean[k][0]).abs()
    if dphi.max().item() > 0:
        idx = torch.nonzero(dphi > 0)
        print("   wrong phi voxels:", idx.shape[0], "first:", idx[:6].tolist(), "last:", idx[-3:].tolist(),
              " per channel:", [int((dphi[c] > 0).sum()) for c in range(3)], flush=True)
        # are the wrong values equal to the OTHER volume's clean phi (stale) ?
        other = clean[1 - k][0]
        m = dphi > 0
        print("   equal to the other volume's value at those voxels:", int((r.phi[m] == other[m]).sum()), "of", int(m.sum()), flush=True)
    if bad["Aw"].max().item() > 0:
        from oai_analysis_2_amd import ops
        idx = torch.nonzero(bad["Aw"] > 0).flatten()
        cl = clean[k][1]
        A_net = ops.resize_trilinear(vols[k][None], (D, H, W))[0]
        low = (D // 2, H // 2, W // 2)
        d1, d2 = cl["d1"].view(3, *low), cl["d2"].view(3, *low)
        c1 = ops.compose(d2, None, out_shape=(D, H, W), shortcut=False)
        cand = {"A(id)": A_net.flatten(), "A(c1)": ops.grid_sample3d(A_net[None], c1)[0].flatten(),
                "A(id+d1(id))": ops.grid_sample3d(A_net[None], ops.compose(d1, None, out_shape=(D, H, W), shortcut=False))[0].flatten(),
                "clean z+1": torch.roll(cl["Aw"], -H * W), "clean z-1": torch.roll(cl["Aw"], H * W), "clean z+2": torch.roll(cl["Aw"], -2 * H * W),
                "clean y+1": torch.roll(cl["Aw"], -W), "clean y-1": torch.roll(cl["Aw"], W), "clean x+16": torch.roll(cl["Aw"], -16), "clean x-16": torch.roll(cl["Aw"], 16),
                "other vol clean": clean[1 - k][1]["Aw"], "zero": torch.zeros_like(cl["Aw"])}
        print("   wrong Aw flat idx:", idx[0].item(), "(z,y,x)=", (idx[0].item() // (H * W), (idx[0].item() // W) % H, idx[0].item() % W))
        print("     wrong:", [round(v, 4) for v in dirty["Aw"][idx].tolist()[:8]])
        print("     clean:", [round(v, 4) for v in cl["Aw"][idx].tolist()[:8]])
        for n, t in cand.items():
            print(f"     max |wrong - {n}| = {(dirty['Aw'][idx] - t[idx]).abs().max().item():.2e}")
        sys.stdout.flush()
    print(f"trial {trial} vol {k}: phi {(r.phi - clean[k][0]).abs().max().item():.2e} | " +
          " ".join(f"{n}:{bad[n].max().item():.1e}({int((bad[n] > 0).sum())})" for n in names), flush=True)
