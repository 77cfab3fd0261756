"""GPU, BASELINE sizes (384x384x160, 160 tiles of 32x128x128; ICON 80x192x192): size-independent properties of the
hot path plus spot checks of individual tiles against the oracle (a whole volume takes the CPU oracle ~7 minutes)."""
import numpy as np
import pytest
import torch

from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.synth import make_smooth_field, make_unet_state_dict, make_volume
from oracle import icon as oicon, seg as oseg

pytestmark = pytest.mark.gpu

SHAPE, TILE, OVL, CROP = (160, 384, 384), (32, 128, 128), (8, 16, 16), (8, 16, 16)


@pytest.fixture(scope="module")
def full():
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = make_unet_state_dict(0)
    eng = UNetEngine(sd)
    vol = make_volume(42, SHAPE)
    v = torch.from_numpy(vol).cuda()
    logits = eng.stitch(eng.segment_tiles(v, TILE, OVL, out_mode=2, crop_zyx=CROP), SHAPE, TILE, OVL, CROP)
    prob = eng.stitch(eng.segment_tiles(v, TILE, OVL, out_mode=0, crop_zyx=CROP), SHAPE, TILE, OVL, CROP)
    return dict(sd=sd, eng=eng, vol=vol, v=v, logits=logits, prob=prob)


def test_full_volume_properties(full):
    eng, v, prob, logits = full["eng"], full["v"], full["prob"], full["logits"]
    assert prob.shape == (2, *SHAPE)
    # the zeroed 8/16/16 frame of Partition.assemble (image_transforms.py:509-513) and nothing but probabilities inside
    p = prob.cpu().numpy()
    assert p[:, :8].max() == 0 and p[:, -8:].max() == 0 and p[:, :, :16].max() == 0 and p[:, :, -16:].max() == 0
    assert p[:, :, :, :16].max() == 0 and p[:, :, :, -16:].max() == 0
    inner = p[:, 8:-8, 16:-16, 16:-16]
    assert inner.min() > 0.0 and inner.max() < 1.0
    # prob == sigmoid(logits) evaluated the same way, mask == (prob > 0.5): the three output modes agree voxel for voxel
    lg = logits[:, 8:-8, 16:-16, 16:-16]
    assert torch.equal(1.0 / (1.0 + torch.exp(-lg)) > 0.5, prob[:, 8:-8, 16:-16, 16:-16] > 0.5) or \
        ((1.0 / (1.0 + torch.exp(-lg)) > 0.5) != (prob[:, 8:-8, 16:-16, 16:-16] > 0.5)).sum() < 10
    mask = eng.stitch(eng.segment_tiles(v, TILE, OVL, out_mode=1, crop_zyx=CROP), SHAPE, TILE, OVL, CROP)
    assert torch.equal(mask, (prob > 0.5).float())
    # determinism: a second run is bit-identical
    again = eng.stitch(eng.segment_tiles(v, TILE, OVL, out_mode=0, crop_zyx=CROP), SHAPE, TILE, OVL, CROP)
    assert torch.equal(again, prob)
    # tile-range sharding over 8 "ranks" (20 tiles each) and other batch sizes give the same maps
    from oai_analysis_2_amd.parallel import tile_range_for_rank
    parts = [eng.segment_tiles(v, TILE, OVL, tile_range_for_rank(160, r, 8), 0, 20, CROP) for r in range(8)]
    assert torch.equal(eng.stitch(torch.cat(parts), SHAPE, TILE, OVL, CROP), prob)
    # ... and the cost-balanced split (border tiles are cheaper) covers the same tiles with equal work per rank
    costs = eng.tile_costs(SHAPE, TILE, OVL, CROP)
    assert len(costs) == 160 and abs(sum(costs) - eng.volume_flops(SHAPE, TILE, OVL, CROP)) < 1e-6 * sum(costs)
    assert min(costs) < 0.75 * max(costs) and costs[0] == min(costs)                  # a corner tile vs an interior tile
    ranges = [tile_range_for_rank(160, r, 8, costs) for r in range(8)]
    work = [sum(costs[b:e]) for b, e in ranges]
    assert max(work) - min(work) <= max(costs) and ranges[0][1] - ranges[0][0] > 20
    parts = [eng.segment_tiles(v, TILE, OVL, rg, 0, 24, CROP) for rg in ranges]
    assert torch.equal(eng.stitch(torch.cat(parts), SHAPE, TILE, OVL, CROP), prob)
    # border-tile trimming off (crop unknown to the segment call) computes more but stitches to the same maps
    untrimmed = eng.stitch(eng.segment_tiles(v, TILE, OVL, out_mode=0, batch=16), SHAPE, TILE, OVL, CROP)
    assert torch.equal(untrimmed, prob)


@pytest.mark.parametrize("tile_index", [0, 37, 90, 159])      # a corner, two interior/edge tiles, the last corner
def test_full_volume_tiles_match_oracle(full, tile_index):
    """Tile t of the volume, cut by the oracle's Partition, through the oracle U-Net == the kept block of the GPU run."""
    g = oseg.tile_geometry(SHAPE, TILE[::-1], OVL[::-1])
    i, j, k = tile_index // 16, (tile_index // 4) % 4, tile_index % 4
    padded = np.pad(full["vol"], [(int(a), int(b)) for a, b in zip(g["pad_lo"], g["pad_hi"])], mode="reflect")
    tile = padded[16 * i:16 * i + 32, 96 * j:96 * j + 128, 96 * k:96 * k + 128]
    ref = oseg.unet_forward(torch.from_numpy(np.ascontiguousarray(tile))[None, None], full["sd"])[0].numpy()
    ref = ref[:, 8:24, 16:112, 16:112]
    got = full["logits"][:, 16 * i:16 * i + 16, 96 * j:96 * j + 96, 96 * k:96 * k + 96].cpu().numpy()
    z0, z1 = (8 if i == 0 else 0), (8 if i == 9 else 16)          # the frame is zeroed: compare what assemble keeps
    y0, y1 = (16 if j == 0 else 0), (80 if j == 3 else 96)
    x0, x1 = (16 if k == 0 else 0), (80 if k == 3 else 96)
    a, b = got[:, z0:z1, y0:y1, x0:x1], ref[:, z0:z1, y0:y1, x0:x1]
    assert np.abs(a - b).max() / np.abs(ref).max() < 1e-4


def test_full_size_resample_identity_and_shift():
    """oai_resample_through_disp at 384x384x160: zero field + same geometry = the input; a constant field = a shift."""
    from oai_analysis_2_amd import ops
    from oai_analysis_2_amd.registration import resample_affines
    vol = make_volume(7, SHAPE)
    img = Image(vol, [0.36, 0.36, 0.7], [3.0, -2.0, 1.0])
    b2n, n2a = resample_affines(img, img, (80, 192, 192))
    zero = torch.zeros((80, 192, 192, 3), dtype=torch.float64, device="cuda")
    out = ops.resample_through_disp(torch.from_numpy(vol).cuda(), zero, b2n, n2a, SHAPE).cpu().numpy()
    assert np.abs(out - vol).max() < 1e-6
    # one network voxel along x = 384/192 = 2 image voxels: out[x] = in[x + 2], zero where x + 2 leaves the buffer
    shift = zero.clone()
    shift[..., 0] = 1.0
    out = ops.resample_through_disp(torch.from_numpy(vol).cuda(), shift, b2n, n2a, SHAPE).cpu().numpy()
    assert np.abs(out[:, :, :-2] - vol[:, :, 2:]).max() < 1e-6
    assert out[:, :, -1].max() == 0.0          # beyond the buffer: ITK's default pixel


def test_full_size_compose_inverse_consistency():
    """phi o phi^-1 ~ id on a synthetic inverse pair at the network resolution (SURVEY 8c invariant)."""
    from oai_analysis_2_amd import ops
    net = (80, 192, 192)
    d = torch.from_numpy(make_smooth_field(11, net, 0.01)).cuda()
    ident = oicon.identity_map(net)[0].cuda()
    phi = ops.compose(d, None, shortcut=True)                       # id + d
    inv = ident - d
    for _ in range(16):                                             # fixed point of inv = x - d(inv), contraction ~0.4
        inv = ident - ops.grid_sample3d(d, inv)
    back = ops.compose(d, inv)                                      # phi(inv(x)) = inv + d(inv)
    interior = (back - ident)[:, 8:-8, 16:-16, 16:-16]              # away from the border clamp of grid_sample
    assert interior.abs().max().item() < 1e-5
    assert torch.equal(phi, ident + d)
