"""Randomised parity of the registration-side kernels vs the oracle: grid_sample / compose on odd shapes, the fused prob-map
resample with rotated / flipped image orientations, image_normalize, one ICON direction on random network shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import ops
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.registration import DisplacementTransform, IconEngine, deform_probmap
from oai_analysis_2_amd.synth import identity_map, make_icon_state_dict, make_smooth_field, make_volume
from oracle import icon as oicon, resample as oresample      # (test helper, not product)

rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

def rot(flip):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if (np.linalg.det(q) < 0) != flip: q[:, 2] *= -1
    return q

# ---- warp / compose
for case in range(8):
    shape = tuple(int(rng.integers(2, 40)) for _ in range(3))
    src = tuple(int(rng.integers(2, 40)) for _ in range(3))
    img, field = make_volume(case, src), make_smooth_field(case + 1, src, 0.05)
    coords = (identity_map(shape) + make_smooth_field(case + 2, shape, 0.2) + rng.normal(size=(3, 1, 1, 1)).astype(np.float32) * 0.3).astype(np.float32)
    ref_w = oicon.sample_at(torch.from_numpy(img)[None, None], torch.from_numpy(coords)[None])[0].numpy()
    ref_c = (torch.from_numpy(coords)[None] + oicon.sample_at(torch.from_numpy(field)[None], torch.from_numpy(coords)[None]))[0].numpy()
    ew = np.abs(ops.grid_sample3d(dev(img[None]), dev(coords)).cpu().numpy() - ref_w).max()
    ec = np.abs(ops.compose(dev(field), dev(coords)).cpu().numpy() - ref_c).max()
    print(f"warp/compose out {shape} src {src}: {ew:.2e} {ec:.2e}")
    assert ew < 3e-6 and ec < 3e-6

# ---- prob-map resample through phi with arbitrary orientations
for case in range(6):
    shA = tuple(int(rng.integers(6, 30)) for _ in range(3)); shB = tuple(int(rng.integers(6, 30)) for _ in range(3))
    net = tuple(int(rng.integers(4, 20)) for _ in range(3))
    A = Image(make_volume(case, shA), rng.uniform(0.3, 1.2, 3), rng.normal(size=3) * 10, rot(bool(case & 1)))
    B = Image(make_volume(case + 9, shB), rng.uniform(0.3, 1.2, 3), A.origin + rng.normal(size=3) * 2, rot(bool(case & 2)) if case % 3 else A.direction)
    phi = (identity_map(net) + make_smooth_field(case, net, 0.05)).astype(np.float32)
    disp = oicon.displacement_itk(torch.from_numpy(phi)[None])
    ref = oresample.resample_through_phi(A.array.astype(np.float64), disp, A, B)
    got = deform_probmap(DisplacementTransform(np.asarray(disp), A, B, phi), A, B, A).array
    err = np.abs(got - ref).max()
    print(f"resample A{shA} B{shB} net{net}: {err:.2e}  (nonzero fraction {float((ref != 0).mean()):.2f})")
    assert err < 2e-5

# ---- image_normalize
for case in range(6):
    n = int(rng.integers(10, 300000))
    x = (rng.normal(size=n) * rng.uniform(0.1, 100) + rng.uniform(-50, 50)).astype(np.float32)
    lo, hi = float(rng.uniform(0, 5)), float(rng.uniform(90, 100))
    from oracle import normalize as onorm
    ref = onorm.image_normalize(x.reshape(1, 1, -1), lo, hi, 0.0, 1.0)
    ref = (ref[0] if isinstance(ref, tuple) else ref).reshape(-1)
    got = ops.image_normalize(dev(x), lo, hi, 0.0, 1.0).cpu().numpy()
    print(f"normalize n={n} pct=({lo:.2f},{hi:.2f}): equal={np.array_equal(got, ref)} maxdiff={np.abs(got - ref).max():.2e}")
    assert np.abs(got - ref).max() < 1e-6

# ---- one ICON direction on random network shapes and step trees (each grid a U-Net runs on needs every axis >= 17)
TREES = ["3step", "4step", "multires", "multires4", ("two", ("down", ("two", ("down", "u"), ("down", "u"))), ("two", "u", "u"))]
for case in range(5):
    tree = TREES[case % len(TREES)]
    deep = tree in ("multires", "multires4") or not isinstance(tree, str)
    lo = 34 if deep else 17
    net = tuple(int(2 * rng.integers(lo, lo + 6)) for _ in range(3))
    sd = make_icon_state_dict(case, 0.1, tree)
    a, b = make_volume(case, net), make_volume(case + 5, net)
    ref = oicon.regis_net_direction(torch.from_numpy(a)[None, None], torch.from_numpy(b)[None, None], sd)[0].numpy()
    eng = IconEngine(sd, net)
    got = eng.phi(dev(a), dev(b)).cpu().numpy()
    d_ref = ref - identity_map(net)
    rel = np.abs(got - ref).max() / np.abs(d_ref).max()
    print(f"ICON net {net} tree {eng.tree.describe()}: displacement rel err {rel:.2e}")
    assert rel < 1e-4
print("all fuzz cases passed")
