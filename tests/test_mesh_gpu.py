"""GPU parity for SURVEY 8f row 3: HIP marching cubes / smoothing / point-to-mesh distance (through the C ABI) vs oracle/mesh.py."""
import numpy as np
import pytest

from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.synth import make_volume
from oracle import mesh as om

pytestmark = pytest.mark.gpu


def _ellipsoid(shape, c, r, squash=(1.0, 1.0, 1.0)):
    z, y, x = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]].astype(np.float32)
    d = np.sqrt(squash[0] * (x - c[2]) ** 2 + squash[1] * (y - c[1]) ** 2 + squash[2] * (z - c[0]) ** 2)
    return (1.0 / (1.0 + np.exp(d - r))).astype(np.float32)


@pytest.mark.parametrize("shape,seed", [((24, 40, 36), 0), ((17, 33, 29), 1), ((2, 2, 2), 2), ((5, 3, 70), 3), ((64, 96, 80), 4)])
def test_marching_cubes_bit_exact(shape, seed):
    """noisy volumes: every one of the 256 cases, surfaces cut by the volume border, ragged sizes -- indices and float32
    vertices identical to the oracle"""
    from oai_analysis_2_amd import mesh_processing as mp
    vol = make_volume(seed, shape) if min(shape) > 2 else np.random.default_rng(seed).random(shape).astype(np.float32)
    level = float(np.median(vol))
    rv, rf = om.marching_cubes(vol, level, (0.36, 0.37, 0.7))
    gv, gf = mp.marching_cubes(vol, level, (0.36, 0.37, 0.7))
    assert gv.shape == rv.shape and gf.shape == rf.shape and len(rf) > 0
    assert np.array_equal(gf, rf)
    assert np.array_equal(gv, rv)


def test_marching_cubes_empty_and_full():
    from oai_analysis_2_amd import mesh_processing as mp
    v, f = mp.marching_cubes(np.zeros((8, 9, 10), np.float32), 0.5)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = mp.marching_cubes(np.ones((8, 9, 10), np.float32), 0.5)
    assert v.shape == (0, 3) and f.shape == (0, 3)


def test_smoothing_matches_oracle():
    from oai_analysis_2_amd import mesh_processing as mp
    v, f = om.marching_cubes(_ellipsoid((30, 34, 32), (14.2, 16.9, 15.5), 9.0, (1.0, 1.3, 0.8)), 0.5, (0.5, 0.5, 0.7))
    ref = om.smooth(v, f, 25, 0.05)
    got = mp.smooth_mesh(mp.Mesh(v, f), num_iterations=25, relaxation_factor=0.05).verts
    assert np.abs(got - ref).max() < 2e-5
    one = mp.smooth_mesh(mp.Mesh(v, f), num_iterations=1, relaxation_factor=0.05).verts        # odd / even ping-pong both land in `out`
    assert np.abs(one - om.smooth(v, f, 1, 0.05)).max() < 4e-6                                # 1-2 ulp at coordinates ~16 (fma contraction)


def test_point_distance_matches_oracle():
    from oai_analysis_2_amd import mesh_processing as mp
    v, f = om.marching_cubes(_ellipsoid((30, 34, 32), (14.2, 16.9, 15.5), 9.0), 0.5)
    rng = np.random.default_rng(5)
    pts = np.concatenate([v[:300] + rng.normal(size=(300, 3)).astype(np.float32) * 2.0,          # near the surface: all Voronoi regions
                          rng.uniform(0, 34, size=(300, 3)).astype(np.float32), v[:50]])
    ref = om.distance_to_mesh(pts, v, f)
    for broad in (False, True):                                                                  # brute force / uniform-grid broad phase
        got = mp.point_distance(pts, mp.Mesh(v, f), broad_phase=broad)
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, ref.max()), broad
        assert np.abs(got[-50:]).max() < 1e-3                                                    # mesh vertices lie on the mesh
    far = (rng.uniform(-200, 200, size=(64, 3))).astype(np.float32)                              # far outside the grid: every ring is walked
    assert np.allclose(mp.point_distance(far, mp.Mesh(v, f), True), mp.point_distance(far, mp.Mesh(v, f), False), rtol=1e-6)


def test_thickness_of_a_shell():
    """get_thickness_mesh end to end (mesh_processing.py:381-395) on a curved plate of known thickness: the inner / outer split
    and both distance directions recover it."""
    from oai_analysis_2_amd import mesh_processing as mp
    D, H, W = 48, 96, 96
    z, y, x = np.mgrid[0:D, 0:H, 0:W].astype(np.float32)
    # a bowl-shaped slab: |r - R| < T/2 around a sphere centred far below, cut to a cap
    R, T = 60.0, 6.0
    r = np.sqrt((x - 48) ** 2 + (z - 24) ** 2 * 4 + (y + 30) ** 2)
    sig = lambda t: 1.0 / (1.0 + np.exp(np.clip(t, -60, 60)))
    prob = sig(2.0 * (np.abs(r - R) - T / 2)) * sig(2.0 * (np.sqrt((x - 48) ** 2 + (z - 24) ** 2 * 4) - 30))       # cap of radius 30 voxels
    img = Image(prob.astype(np.float32), [1.0, 1.0, 1.0])
    mesh = mp.get_mesh(img, num_iterations=20, min_cells=100)
    assert mesh.GetNumberOfCells() > 3000
    inner, outer = mp.split_mesh(mesh, "TC")
    assert inner.GetNumberOfCells() > 500 and outer.GetNumberOfCells() > 500
    d_in, d_out = mp.get_distance(inner, outer)
    med = np.median(d_in.point_data["Distance"])
    assert abs(med - T) < 1.5, med                                   # plate thickness, up to smoothing and the rim
    assert d_out.point_data["Distance"].shape == (outer.GetNumberOfPoints(),)
    # the Dask task body (dask_processing.py:114-122) is the same chain with the reference's defaults
    from oai_analysis_2_amd.dask_processing import get_thickness
    inner_d = get_thickness(img, "TC")
    assert abs(np.median(inner_d.point_data["Distance"]) - T) < 1.5 and inner_d.GetNumberOfPoints() > 500
