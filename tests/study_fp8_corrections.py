"""Numerical study for DESIGN.md section 6 ("the one idea that removes MFMA work"): what happens to the logits of the reference's golden tile
if the two correction passes of the fp16x3 arithmetic take fp8 operands?   (Run by hand: `python tests/study_fp8_corrections.py`; CPU only,
~10 minutes; lives under tests/ because it checks against the reference-made golden vector `tests/golden/unet_fulltile.npz`.)

Today (fp16x3):  a = a0 + a1, b = b0 + b1 in fp16;  a*b ~= a0*b0 + a0*b1 + a1*b0   (three fp16 MFMA passes, fp32 accumulate).
Variant:         a0*b0 as today;  a0*b1 -> q(a0)*q(b1),  a1*b0 -> q(a1)*q(b0)  with q = fp8 rounding (e4m3 with an ideal power-of-two
                 scale per tensor, as a scaled MFMA would apply it, or e5m2 unscaled).  On gfx950 the fp8 MFMA runs at twice the fp16
                 rate, so the three passes would cost two.
Emulation: every pass is an fp64 convolution of the (exactly representable) operand tensors, i.e. exact products and a near-exact sum;
what is measured is the effect of the operand formats alone."""
import os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle.seg import ENCODER, UPCONV, DECONV3

torch.set_num_threads(os.cpu_count() or 8)


def split16(x):
    h = x.to(torch.float16).to(torch.float32)
    l = (x - h).to(torch.float16).to(torch.float32)
    return h, l


def q_e4m3_scaled(x):            # ideal per-tensor power-of-two scale into e4m3's range (max 448), then back
    m = float(x.abs().max())
    if m == 0.0: return x
    s = 2.0 ** np.floor(np.log2(256.0 / m))
    return (x * s).to(torch.float8_e4m3fn).to(torch.float32) / s


def q_e5m2(x):
    return x.to(torch.float8_e5m2).to(torch.float32)


def conv(name, x, w):
    x, w = x.double(), w.double()
    if name in ENCODER: return F.conv3d(x, w, None, stride=1, padding=1)
    if name in UPCONV: return F.conv_transpose3d(x, w, None, stride=2, padding=0)
    return F.conv_transpose3d(x, w, None, stride=1, padding=1)


def block(mode, name, x, sd):
    w, b = sd[f"{name}.0.weight"], sd[f"{name}.0.bias"]
    if mode == "fp32" or name == "ec0":                    # (ec0 is exact fp32 FMA in the library)
        y = conv(name, x, w)
    else:
        a0, a1 = split16(x); b0, b1 = split16(w)
        y = conv(name, a0, b0)
        if mode == "fp16x3": y = y + conv(name, a0, b1) + conv(name, a1, b0)
        elif mode == "fp16x2": y = y + conv(name, a0, b1)                         # (dropping a1*b0: the documented 2.4e-4 case)
        else:
            q = q_e4m3_scaled if mode == "e4m3" else q_e5m2
            y = y + conv(name, q(a0), q(b1)) + conv(name, q(a1), q(b0))
    return F.relu(y + b.double().view(1, -1, 1, 1, 1)).float()


@torch.no_grad()
def forward(mode, x, sd):
    B = lambda n, t: block(mode, n, t, sd)
    e0 = B("ec0", x); syn0 = B("ec1", e0); e2 = B("ec2", F.max_pool3d(syn0, 2)); syn1 = B("ec3", e2)
    e4 = B("ec4", F.max_pool3d(syn1, 2)); syn2 = B("ec5", e4); e6 = B("ec6", F.max_pool3d(syn2, 2)); e7 = B("ec7", e6)
    d8 = B("dc8", torch.cat((B("dc9", e7), syn2), 1)); d7 = B("dc7", d8)
    d5 = B("dc5", torch.cat((B("dc6", d7), syn1), 1)); d4 = B("dc4", d5)
    d2 = B("dc2", torch.cat((B("dc3", d4), syn0), 1)); d1 = B("dc1", d2)
    return F.conv3d(d1.double(), sd["dc0.weight"].double(), sd["dc0.bias"].double()).float()


if __name__ == "__main__":
    z = np.load(os.path.join(ROOT, "tests", "golden", "unet_fulltile.npz"))
    sd = {k: v.float() for k, v in make_unet_state_dict(seed=int(z["weight_seed"])).items()}
    x = torch.from_numpy(make_volume(int(z["volume_seed"]), (32, 128, 128)))[None, None]
    ref = z["logits_centre"]; scale = float(z["logits_abs_max"])
    pref = 1.0 / (1.0 + np.exp(-ref.astype(np.float64)))
    print(f"golden tile: reference logits |max| {scale:.3f}; error of each operand format against the reference's fp32 run:")
    for mode in (sys.argv[1:] or ["fp32", "fp16x3", "e4m3", "e5m2", "fp16x2"]):
        t = time.time()
        y = forward(mode, x, sd)[0, :, 8:24, 16:112, 16:112].numpy()
        p = 1.0 / (1.0 + np.exp(-y.astype(np.float64)))
        flips = int(((p > 0.5) != (pref > 0.5)).sum())
        print(f"  {mode:7s} max|dlogit| / max|logit| = {np.abs(y - ref).max() / scale:.2e}   sum|dp| over the kept centre "
              f"({ref[0].size} voxels x 2) = {np.abs(p - pref).sum():.3f} (= {np.abs(p - pref).sum() / 2 * 23592960 / ref[0].size:.1f} per 23.6 M voxels; budget 12)   "
              f"mask flips {flips}   [{time.time() - t:.0f} s]", flush=True)
