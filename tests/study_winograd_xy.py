"""Study (CPU, torch; not collected by pytest -- run `python tests/study_winograd_xy.py`): would the y axis of SOME layers in Winograd F(2,3) form, on top of
the shipped x axis, keep the whole network inside this repo's full-size gates?  DESIGN.md section 6 names it as the one step left that removes matrix work
(12 instead of 18 MFMA-taps per output); `scripts/study/winograd_xy_error.py` prices one layer (up to 1.9 x the direct form's error, x-only 1.45 x).  Here the
fp16x3 arithmetic -- activations and weights as fp16 pairs, products a0 b0 + a0 b1 + a1 b0, fp32 accumulation, activations stored as fp16 pairs behind a
calibration exponent -- is emulated layer by layer through the reference network (networks.py:109-149, oracle/seg.py) on one 16 x 64 x 64 tile, with the plain
k3 layers (ec3-ec7, dc8, dc7, dc5, dc4, dc2) in direct / x / x + y form, and compared with the float64 run of the same network the way
tests/test_fullsize_gpu.py does: distance from the float64 maps relative to the fp32 network's own distance from them (gate 2.2 x).  The summation ORDER inside
a conv is torch's, not the kernels': this ranks the forms, it does not predict the third digit."""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle import seg as oseg

torch.set_num_threads(8)
WINO = ("ec3", "ec4", "ec5", "ec6", "ec7", "dc8", "dc7", "dc5", "dc4", "dc2")
T3 = ("dc8", "dc7", "dc5", "dc4", "dc2", "dc1")
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)

def split(x):
    h0 = x.half().float()
    return h0, (x - h0).half().float()

def quant(x):                                   # an activation as format S stores it: fp16 pair behind a power-of-two exponent (max in [2^10, 2^11))
    m = float(x.abs().max())
    if m == 0.0: return x, 1.0
    e = 2.0 ** (10 - np.floor(np.log2(m)))
    h0, h1 = split(x * e)
    return (h0 + h1) / e, e

def conv3(a, w):                                # three-pass product, fp32 accumulation; a [1,C,...] already scaled, w [Co,Ci,kz,ky,kx] already scaled (float32 tensors)
    a0, a1 = split(a); b0, b1 = split(w)
    return (F.conv3d(a0, b0) + F.conv3d(a0, b1)) + F.conv3d(a1, b0)

def layer(x, w, b, form):
    """relu(conv3d(x, w, pad 1) + b) in emulated fp16x3; w as a plain conv weight [Co,Ci,3,3,3] (float32)."""
    xq, e = quant(x)
    xs = F.pad(xq * e, (1, 1, 1, 1, 1, 1))
    ws = 2.0 ** (8 - torch.ceil(torch.log2(w.abs().amax(dim=(1, 2, 3, 4)))))             # per-cout power of two
    wsc = (w * ws.view(-1, 1, 1, 1, 1)).double()
    D, H, W = x.shape[2:]
    if form == "direct":
        y = conv3(xs, wsc.float())
    elif form == "x":
        P = W // 2
        d = [xs[..., k:k + 2 * P:2] for k in range(4)]
        t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
        u = torch.einsum("fx,oizyx->foizy", G, wsc)                                       # [4, Co, Ci, 3, 3]
        m = [conv3(t[f], u[f].unsqueeze(-1).float()) for f in range(4)]
        y = torch.stack([(m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]], dim=-1).reshape(1, -1, D, H, W)
    else:
        P, Q = W // 2, H // 2
        d = [[xs[..., ky:ky + 2 * Q:2, kx:kx + 2 * P:2] for kx in range(4)] for ky in range(4)]
        ty = [[d[0][k] - d[2][k] for k in range(4)], [d[1][k] + d[2][k] for k in range(4)], [d[2][k] - d[1][k] for k in range(4)], [d[1][k] - d[3][k] for k in range(4)]]
        t = [[a[0] - a[2], a[1] + a[2], a[2] - a[1], a[1] - a[3]] for a in ty]            # t[fy][fx], fp32 adds
        u = torch.einsum("ay,oizyx,bx->aboiz", G, wsc, G)                                 # [4, 4, Co, Ci, 3]
        m = [[conv3(t[fy][fx], u[fy, fx].unsqueeze(-1).unsqueeze(-1).float()) for fx in range(4)] for fy in range(4)]
        my = [[(m[0][k] + m[1][k]) + m[2][k] for k in range(4)], [(m[1][k] - m[2][k]) - m[3][k] for k in range(4)]]
        rows = [torch.stack([(a[0] + a[1]) + a[2], (a[1] - a[2]) - a[3]], dim=-1) for a in my]      # each [1, Co, D, Q, P, 2]
        y = torch.stack(rows, dim=4).reshape(1, -1, D, H, W)                              # [1, Co, D, Q, 2, P, 2]
    y = y / (ws.view(1, -1, 1, 1, 1) * e)
    return torch.relu(y + b.view(1, -1, 1, 1, 1))

def as_conv(sd, name):
    w = sd[f"{name}.0.weight"]
    return w.flip(2, 3, 4).transpose(0, 1).contiguous() if name in T3 else w

def network(x, sd, forms):
    """The reference network with every k3 layer emulated (ec0 and the k2 up-convs / the head in plain fp32 on the stored activations)."""
    def blk(x, name):
        w, b = as_conv(sd, name), sd[f"{name}.0.bias"]
        if name == "ec0": return quant(torch.relu(F.conv3d(x, w, b, padding=1)))[0]
        return layer(x, w, b, forms.get(name, "direct"))
    def up(x, name):
        return quant(torch.relu(F.conv_transpose3d(quant(x)[0], sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], stride=2)))[0]
    e0 = blk(x, "ec0"); syn0 = blk(e0, "ec1"); e1 = F.max_pool3d(quant(syn0)[0], 2)
    e2 = blk(e1, "ec2"); syn1 = blk(e2, "ec3"); e3 = F.max_pool3d(quant(syn1)[0], 2)
    e4 = blk(e3, "ec4"); syn2 = blk(e4, "ec5"); e5 = F.max_pool3d(quant(syn2)[0], 2)
    e7 = blk(blk(e5, "ec6"), "ec7")
    d7 = blk(blk(torch.cat((up(e7, "dc9"), quant(syn2)[0]), 1), "dc8"), "dc7")
    d4 = blk(blk(torch.cat((up(d7, "dc6"), quant(syn1)[0]), 1), "dc5"), "dc4")
    d1 = blk(blk(torch.cat((up(d4, "dc3"), quant(syn0)[0]), 1), "dc2"), "dc1")
    return F.conv3d(quant(d1)[0], sd["dc0.weight"], sd["dc0.bias"])

if __name__ == "__main__":
    shape = (16, 64, 64)
    for seed in (0, 3):
        sd = make_unet_state_dict(seed=seed)
        x = torch.from_numpy(make_volume(40 + seed, shape))[None, None]
        truth = torch.sigmoid(oseg.unet_forward(x.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}))
        ref = torch.sigmoid(oseg.unet_forward(x, sd)).double()
        base = (ref - truth).abs().sum().item()
        print(f"weights seed {seed}: the fp32 network's own distance from its float64 run: sum|dp| = {base:.3e} over {truth.numel()} values")
        variants = [("direct everywhere", {}), ("x on the ten plain layers (shipped)", {k: "x" for k in WINO}),
                    ("+ y on dc2", {**{k: "x" for k in WINO}, "dc2": "xy"}),
                    ("+ y on dc2, dc5, dc8", {**{k: "x" for k in WINO}, "dc2": "xy", "dc5": "xy", "dc8": "xy"}),
                    ("x + y on all ten", {k: "xy" for k in WINO})]
        for name, forms in variants:
            t0 = time.time()
            p = torch.sigmoid(network(x, sd, forms)).double()
            print(f"  fp16x3, {name:40s} distance from float64 {(p - truth).abs().sum().item() / base:5.2f} x the fp32 network's   (from the fp32 network: {(p - ref).abs().sum().item() / base:5.2f} x)   [{time.time() - t0:.0f} s]")
