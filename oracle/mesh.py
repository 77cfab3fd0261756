"""Oracle for the step AFTER the hot path: iso-surface, smoothing, closest-point distance.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**: the reference computes this with
``skimage.measure.marching_cubes`` (Lewiner), ``vtkSmoothPolyDataFilter`` and ``vtkDistancePolyDataFilter``
(oai_analysis/mesh_processing.py:298-340, 381-395); none of skimage / vtk / trimesh / itk is installed here and the reference's
tests assert nothing on meshes, so this file restates the published algorithms:

* marching cubes (Lorensen & Cline) with a case table GENERATED from first principles (``mc_table``): on every cube face the
  iso-line segments are fixed by the face's corner signs alone (ambiguous faces always isolate the inside corners), so two
  cubes sharing a face agree and the surface is watertight; the segments chain into closed loops, each fan-triangulated from
  the first loop vertex (lowest edge first) that draws no diagonal inside a cube face, so the surface is also an oriented
  2-manifold (every directed edge once).  The vertex SET equals that of any marching-cubes variant (one vertex per sign-changing grid edge, linear
  interpolation); the triangulation differs from Lewiner's in ambiguous cells.
* Laplacian smoothing x <- x + f * (mean of edge neighbours - x), Jacobi sweeps (vtk sweeps in place, Gauss-Seidel order).
* unsigned distance from points to a triangle soup, exact closest point on each triangle (Ericson, Real-Time Collision
  Detection 5.1.5), brute force.

Conventions: volume [z][y][x]; corner c of a cell sits at offset (c & 1, (c >> 1) & 1, c >> 2) in (x, y, z); edge id =
axis * 4 + j (axis 0 = x: j = cy + 2 cz; axis 1 = y: j = cx + 2 cz; axis 2 = z: j = cx + 2 cy); "inside" = value > iso;
triangles wind so that normals point from inside to outside; vertices are (x, y, z) * spacing, float32.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import numpy as np

# faces of the unit cube as corner quadruples, counter-clockwise when seen from OUTSIDE the cube
_FACES = (
    (0, 4, 6, 2),   # x = 0
    (1, 3, 7, 5),   # x = 1
    (0, 1, 5, 4),   # y = 0
    (2, 6, 7, 3),   # y = 1
    (0, 2, 3, 1),   # z = 0
    (4, 5, 7, 6),   # z = 1
)


def corner_offset(c: int) -> Tuple[int, int, int]:
    return c & 1, (c >> 1) & 1, c >> 2


def edge_of(c0: int, c1: int) -> int:
    """id of the cube edge joining two adjacent corners"""
    d = c0 ^ c1
    lo = min(c0, c1)
    x, y, z = corner_offset(lo)
    if d == 1:
        return 0 * 4 + y + 2 * z
    if d == 2:
        return 1 * 4 + x + 2 * z
    if d == 4:
        return 2 * 4 + x + 2 * y
    raise ValueError("corners are not adjacent")


def edge_corners(e: int) -> Tuple[int, int]:
    axis, j = divmod(e, 4)
    a, b = j & 1, j >> 1
    lo = {0: (a << 1) | (b << 2), 1: a | (b << 2), 2: a | (b << 1)}[axis]
    return lo, lo | (1 << axis)


def _edge_faces(e: int):
    """the two cube faces (axis, side) that contain cube edge e"""
    axis, j = divmod(e, 4)
    others = [ax for ax in range(3) if ax != axis]
    return {(others[0], j & 1), (others[1], j >> 1)}


def _share_face(e1: int, e2: int) -> bool:
    return bool(_edge_faces(e1) & _edge_faces(e2))


@lru_cache(maxsize=None)
def mc_table() -> np.ndarray:
    """[256][16] int8: up to 5 triangles as edge-id triples, -1 terminated.  Bit c of the case index = corner c inside."""
    table = -np.ones((256, 16), dtype=np.int8)
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        nxt = {}
        for quad in _FACES:
            enters, exits = [], []                       # positions k: crossing on the face edge quad[k] -> quad[k+1]
            for k in range(4):
                a, b = quad[k], quad[(k + 1) % 4]
                if inside[a] and not inside[b]:
                    exits.append(k)
                elif not inside[a] and inside[b]:
                    enters.append(k)
            for ke in enters:
                # the inside arc that starts at this enter crossing ends at the first exit after it (ccw): on an ambiguous face
                # that isolates each inside corner.  Directed enter -> exit, which winds the loops outward (checked below).
                kx = min(exits, key=lambda k: (k - ke) % 4)
                e_from = edge_of(quad[ke], quad[(ke + 1) % 4])
                e_to = edge_of(quad[kx], quad[(kx + 1) % 4])
                assert e_from not in nxt
                nxt[e_from] = e_to
        tris = []
        seen = set()
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, e = [], start
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            assert e == start and len(loop) >= 3
            # fan apex: the first rotation of the loop with no diagonal lying IN a cube face -- the neighbouring cube could
            # draw the same diagonal on the shared face, and the edge would carry four triangles (non-manifold)
            n = len(loop)
            rot = min(range(n), key=lambda r: (sum(_share_face(loop[r], loop[(r + i) % n]) for i in range(2, n - 1)), r))
            assert sum(_share_face(loop[rot], loop[(rot + i) % n]) for i in range(2, n - 1)) == 0
            loop = loop[rot:] + loop[:rot]
            for i in range(1, len(loop) - 1):
                tris.append((loop[0], loop[i], loop[i + 1]))
        assert len(tris) <= 5, (case, tris)
        flat = [e for t in tris for e in t]
        table[case, :len(flat)] = flat
    # orientation check on the single-corner case: corner 0 inside -> normal (+,+,+)
    p = {e: np.mean([corner_offset(c) for c in edge_corners(e)], axis=0) for e in range(12)}
    t = table[1, :3]
    n = np.cross(p[t[1]] - p[t[0]], p[t[2]] - p[t[0]])
    assert (n > 0).all(), n
    return table


def marching_cubes(vol: np.ndarray, iso: float = 0.5, spacing=(1.0, 1.0, 1.0)) -> Tuple[np.ndarray, np.ndarray]:
    """(verts float32 [n,3] in (x,y,z)*spacing, faces int32 [m,3]).  Vertex order: grid edges in (z, y, x, axis) order;
    triangle order: cells in (z, y, x) order, table order inside a cell."""
    v = np.ascontiguousarray(vol, dtype=np.float32)
    D, H, W = v.shape
    iso = np.float32(iso)
    ins = v > iso
    # ---- vertices: one per sign-changing grid edge, owned by the lower voxel
    flag = np.zeros((D, H, W, 3), dtype=bool)
    flag[:, :, :-1, 0] = ins[:, :, :-1] != ins[:, :, 1:]
    flag[:, :-1, :, 1] = ins[:, :-1, :] != ins[:, 1:, :]
    flag[:-1, :, :, 2] = ins[:-1, :, :] != ins[1:, :, :]
    vid = np.cumsum(flag.reshape(-1), dtype=np.int64) - 1
    vid = np.where(flag.reshape(-1), vid, -1).reshape(D, H, W, 3)
    z, y, x, ax = np.nonzero(flag)
    va = v[z, y, x]
    vb = v[z + (ax == 2), y + (ax == 1), x + (ax == 0)]
    t = (iso - va) / (vb - va)
    pos = np.stack([x, y, z], axis=1).astype(np.float32)
    pos[np.arange(len(ax)), ax] += t
    verts = pos * np.asarray(spacing, dtype=np.float32)[None, :]
    # ---- triangles
    case = np.zeros((D - 1, H - 1, W - 1), dtype=np.int32)
    for c in range(8):
        cx, cy, cz = corner_offset(c)
        case |= ins[cz:D - 1 + cz, cy:H - 1 + cy, cx:W - 1 + cx].astype(np.int32) << c
    table = mc_table().astype(np.int32)
    ntri = (table >= 0).sum(axis=1) // 3
    cz, cy, cx = np.nonzero(ntri[case] > 0)
    faces = []
    cc = case[cz, cy, cx]
    for k in range(5):
        sel = ntri[cc] > k
        if not sel.any():
            break
        tri = np.empty((int(sel.sum()), 3), dtype=np.int64)
        for j in range(3):
            e = table[cc[sel], 3 * k + j]
            axis, jj = e // 4, e % 4
            a, b = jj & 1, jj >> 1
            ox = np.where(axis == 0, 0, a)
            oy = np.where(axis == 0, a, np.where(axis == 1, 0, b))
            oz = np.where(axis == 2, 0, b)
            tri[:, j] = vid[cz[sel] + oz, cy[sel] + oy, cx[sel] + ox, axis]
        order = np.flatnonzero(sel)
        faces.append((order, k, tri))
    if not faces:
        return verts.astype(np.float32), np.zeros((0, 3), np.int32)
    # interleave: cell order first, then k
    idx = np.concatenate([o * 8 + k for o, k, _ in faces])
    tri = np.concatenate([t for _, _, t in faces])
    tri = tri[np.argsort(idx, kind="stable")]
    assert (tri >= 0).all()
    return verts.astype(np.float32), tri.astype(np.int32)


def vertex_adjacency(n_verts: int, faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """CSR (offsets [n+1], neighbours) of the edge graph, neighbours sorted ascending, no duplicates."""
    f = np.asarray(faces, dtype=np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1]])
    key = np.unique(e[:, 0] * n_verts + e[:, 1])
    src, dst = key // n_verts, key % n_verts
    off = np.zeros(n_verts + 1, dtype=np.int64)
    np.add.at(off, src + 1, 1)
    return np.cumsum(off).astype(np.int32), dst.astype(np.int32)


def smooth(verts: np.ndarray, faces: np.ndarray, iterations: int = 150, relaxation: float = 0.01) -> np.ndarray:
    """Jacobi Laplacian smoothing in float32 (vtkSmoothPolyDataFilter defaults: relaxation 0.01, boundary smoothing on)."""
    off, nbr = vertex_adjacency(len(verts), faces)
    deg = np.diff(off).astype(np.float32)
    x = verts.astype(np.float32).copy()
    rows = np.repeat(np.arange(len(verts)), np.diff(off))
    f = np.float32(relaxation)
    for _ in range(iterations):
        s = np.zeros_like(x)
        np.add.at(s, rows, x[nbr])
        mean = s / np.maximum(deg, 1)[:, None]
        x = np.where(deg[:, None] > 0, x + f * (mean - x), x).astype(np.float32)
    return x


def point_triangle_distance(p: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """distance from points p[n,3] to triangles (a,b,c)[m,3] -> [n,m] (Ericson 5.1.5, float64)"""
    p = p[:, None, :].astype(np.float64)
    a, b, c = (t[None].astype(np.float64) for t in (a, b, c))
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        closest = np.empty(np.broadcast_shapes(p.shape, a.shape))
        denom = va + vb + vc
        v = vb / denom
        w = vc / denom
        closest[:] = a + ab * v[..., None] + ac * w[..., None]                     # interior
        m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)                          # edge bc
        ww = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        closest = np.where(m[..., None], b + (c - b) * ww[..., None], closest)
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)                                        # edge ac
        closest = np.where(m[..., None], a + ac * (d2 / (d2 - d6))[..., None], closest)
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)                                        # edge ab
        closest = np.where(m[..., None], a + ab * (d1 / (d1 - d3))[..., None], closest)
        closest = np.where(((d6 >= 0) & (d5 <= d6))[..., None], np.broadcast_to(c, closest.shape), closest)   # vertex c
        closest = np.where(((d3 >= 0) & (d4 <= d3))[..., None], np.broadcast_to(b, closest.shape), closest)   # vertex b
        closest = np.where(((d1 <= 0) & (d2 <= 0))[..., None], np.broadcast_to(a, closest.shape), closest)    # vertex a
    return np.sqrt(((p - closest) ** 2).sum(-1))


def distance_to_mesh(points: np.ndarray, verts: np.ndarray, faces: np.ndarray, chunk: int = 256) -> np.ndarray:
    """unsigned distance of each point to the triangle mesh (vtkDistancePolyDataFilter, SignedDistanceOff)"""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    out = np.empty(len(points), dtype=np.float64)
    for s in range(0, len(points), chunk):
        out[s:s + chunk] = point_triangle_distance(points[s:s + chunk], a, b, c).min(axis=1)
    return out
