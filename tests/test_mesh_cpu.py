"""SURVEY 8f row 3 on the CPU: the marching-cubes case table (oracle generator vs the library's own generator, which is host
code) and the oracle's iso-surface on an analytic shape."""
import ctypes as C

import numpy as np

from oracle import mesh as om


def _ellipsoid(shape=(40, 48, 44), r=12.5):
    z, y, x = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]].astype(np.float32)
    d = np.sqrt((x - 21.3) ** 2 + 1.1 * (y - 23.1) ** 2 + (z - 19.7) ** 2)
    return (1.0 / (1.0 + np.exp(d - r))).astype(np.float32)


def test_case_table_properties():
    t = om.mc_table().astype(np.int32)
    n = (t >= 0).sum(axis=1)
    assert (n % 3 == 0).all() and n.max() == 15 and n[0] == 0 and n[255] == 0
    for case in range(256):
        used = set(t[case][t[case] >= 0].tolist())
        crossing = {e for e in range(12) if ((case >> om.edge_corners(e)[0]) & 1) != ((case >> om.edge_corners(e)[1]) & 1)}
        assert used == crossing                                    # every sign-changing edge carries exactly the vertices used
        # complementary cases use the same vertices
        assert set(t[255 - case][t[255 - case] >= 0].tolist()) == crossing


def test_library_table_equals_oracle_table():
    from oai_analysis_2_amd import _lib
    lib = _lib.load()
    buf = (C.c_byte * (256 * 16))()
    assert lib.oai_mc_table(buf) == 0
    got = np.frombuffer(buf, dtype=np.int8).reshape(256, 16)
    assert np.array_equal(got, om.mc_table())


def test_oracle_surface_is_watertight_and_outward():
    sp = (0.5, 0.6, 0.7)
    v, f = om.marching_cubes(_ellipsoid(), 0.5, sp)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
    key, rkey = e[:, 0] * len(v) + e[:, 1], e[:, 1] * len(v) + e[:, 0]
    assert len(np.unique(key)) == len(key) and np.isin(rkey, key).all()             # closed, consistently oriented 2-manifold
    assert len(v) - len(key) // 2 + len(f) == 2                                     # a sphere
    a, b, c = (v[f[:, k]].astype(np.float64) for k in range(3))
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6
    expect = 4 / 3 * np.pi * 12.5 ** 3 / np.sqrt(1.1) * np.prod(sp)
    assert vol > 0 and abs(vol - expect) / expect < 0.01                            # normals point out of the object


def test_oracle_smoothing_and_distance():
    v, f = om.marching_cubes(_ellipsoid((24, 26, 28), 8.0), 0.5)
    vs = om.smooth(v, f, 10, 0.1)
    assert vs.shape == v.shape and 0 < np.abs(vs - v).max() < 1.0
    # distance from points pushed outward along the radial direction of a big sphere ~ the push length
    c = v.mean(axis=0)
    out = c + (v[:40] - c) * 1.25
    d = om.distance_to_mesh(out, v, f)
    r = np.linalg.norm(v[:40] - c, axis=1)
    assert np.all(d <= 0.25 * r + 1e-6) and np.all(d > 0.15 * r)
    assert np.allclose(om.distance_to_mesh(v[:10], v, f), 0, atol=1e-5)


def test_oracle_surface_of_noise_is_an_oriented_manifold():
    """every one of the 256 cases, ambiguous faces included: each directed edge exactly once, its reverse present unless the
    surface is cut by the volume border (no fan diagonal lies in a cube face, so neighbouring cubes never draw the same edge)"""
    vol = np.random.default_rng(0).random((14, 15, 16)).astype(np.float32)
    v, f = om.marching_cubes(vol, 0.5)
    assert len(np.unique(om.mc_table()[:, 0])) > 1 and len(f) > 5000
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
    key, rkey = e[:, 0] * len(v) + e[:, 1], e[:, 1] * len(v) + e[:, 0]
    assert len(np.unique(key)) == len(key)
    interior = np.all((v[e] > 0.0) & (v[e] < np.array(vol.shape[::-1], np.float32) - 1.0), axis=(1, 2))
    assert np.isin(rkey[interior], key).all()
