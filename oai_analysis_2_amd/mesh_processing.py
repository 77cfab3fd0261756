"""Mesh / thickness step after the hot path -- the surface of oai_analysis/mesh_processing.py on the MI355X.

Reference functions mirrored (same names, argument meaning and return roles):

    get_mesh(itk_image, num_iterations=150)            mesh_processing.py:325-340   marching cubes @0.5 + smoothing
    smooth_mesh(mesh, num_iterations=150)              :298-307
    split_mesh(mesh, mesh_type="FC")                   :353-378   inner / outer surface (KMeans on centroids + normals)
    get_distance(inner_mesh, outer_mesh)               :310-322   closest-point distance, both directions
    get_thickness_mesh(itk_image, mesh_type, ...)      :381-395
    get_cell_centroid / get_cell_normals               :26-46

vtk / trimesh / skimage are not installed here, so meshes are ``Mesh`` objects (float32 vertices [n,3] in (x,y,z)*spacing,
int32 faces [m,3], per-point data) instead of ``vtkPolyData``; ``Mesh.to_vtk()`` adapts when vtk imports.  The three heavy
steps run in HIP kernels behind the C ABI (oai_mc_*, oai_mesh_smooth, oai_mesh_point_distance; csrc/mesh.hip); the edge
graph, the connected-component filter (> 3000 cells, :119-137) and the KMeans split are host logic exactly as in the
reference (sklearn is the reference's own dependency).  Parity is unpinned: see oracle/mesh.py for what is
restated.  There is no CPU fallback for the kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .image import as_image


@dataclass
class Mesh:
    verts: np.ndarray                       # float32 [n,3], (x,y,z) in the image's spacing units
    faces: np.ndarray                       # int32 [m,3]
    point_data: Dict[str, np.ndarray] = field(default_factory=dict)

    def GetNumberOfPoints(self) -> int:      # the vtkPolyData calls the reference makes on meshes
        return len(self.verts)

    def GetNumberOfCells(self) -> int:
        return len(self.faces)

    def GetBounds(self):
        lo, hi = self.verts.min(axis=0), self.verts.max(axis=0)
        return (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])

    def to_vtk(self):  # pragma: no cover - vtk is absent in this environment
        import vtk
        from vtk.util import numpy_support as ns
        cells = vtk.vtkCellArray()
        cells.SetData(ns.numpy_to_vtk(np.arange(0, 3 * len(self.faces) + 1, 3).astype("int")), ns.numpy_to_vtk(self.faces.reshape(-1).astype("int")))
        pts = vtk.vtkPoints()
        pts.SetData(ns.numpy_to_vtk(self.verts.astype(np.float64), deep=True))
        out = vtk.vtkPolyData()
        out.SetPoints(pts)
        out.SetPolys(cells)
        for name, arr in self.point_data.items():
            a = ns.numpy_to_vtk(np.asarray(arr, np.float64), deep=True)
            a.SetName(name)
            out.GetPointData().AddArray(a)
        return out


def _dev(a: np.ndarray, dtype) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise RuntimeError("oai_analysis_2_amd.mesh_processing runs on the GPU only (no CPU fallback)")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- marching cubes ----------------------------------------------------------------------------------------------------------
def marching_cubes(volume_zyx, level: float = 0.5, spacing_xyz=(1.0, 1.0, 1.0)) -> Tuple[np.ndarray, np.ndarray]:
    """(verts, faces) of the iso-surface; ``volume_zyx`` may be a numpy array or a float32 torch tensor already on the device."""
    lib = _lib.load()
    vol = volume_zyx if isinstance(volume_zyx, torch.Tensor) else _dev(np.asarray(volume_zyx), np.float32)
    vol = vol.to(torch.float32).contiguous()
    if not vol.is_cuda:
        vol = vol.cuda()
    D, H, W = (int(v) for v in vol.shape)
    ws = torch.empty(int(lib.oai_mc_workspace_bytes(D, H, W)), dtype=torch.uint8, device=vol.device)
    nv, nt = C.c_longlong(), C.c_longlong()
    with torch.cuda.device(vol.device):
        _lib.check(lib.oai_mc_count(vol.data_ptr(), D, H, W, float(level), ws.data_ptr(), ws.numel(), C.byref(nv), C.byref(nt), _stream()),
                   "oai_mc_count")
        verts = torch.empty((nv.value, 3), dtype=torch.float32, device=vol.device)
        faces = torch.empty((nt.value, 3), dtype=torch.int32, device=vol.device)
        sp = (C.c_float * 3)(*[float(v) for v in spacing_xyz])
        _lib.check(lib.oai_mc_emit(vol.data_ptr(), D, H, W, float(level), sp, ws.data_ptr(), verts.data_ptr(), faces.data_ptr(), _stream()),
                   "oai_mc_emit")
    return verts.cpu().numpy(), faces.cpu().numpy()


# ---- host-side graph helpers -------------------------------------------------------------------------------------------------
def vertex_adjacency(n_verts: int, faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """CSR edge graph (offsets [n+1], neighbours ascending, no duplicates)."""
    f = np.asarray(faces, dtype=np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1]])
    key = np.unique(e[:, 0] * n_verts + e[:, 1])
    src, dst = key // n_verts, key % n_verts
    off = np.zeros(n_verts + 1, dtype=np.int64)
    np.add.at(off, src + 1, 1)
    return np.cumsum(off).astype(np.int32), dst.astype(np.int32)


def keep_large_regions(verts: np.ndarray, faces: np.ndarray, min_cells: int = 3000) -> Tuple[np.ndarray, np.ndarray]:
    """get_vtk_mesh's vtkPolyDataConnectivityFilter loop (mesh_processing.py:114-141): keep connected regions with more than
    ``min_cells`` triangles, drop unreferenced points."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    n = len(verts)
    if len(faces) == 0:
        return verts[:0], faces
    f = faces.astype(np.int64)
    g = coo_matrix((np.ones(2 * len(f), np.int8), (np.concatenate([f[:, 0], f[:, 1]]), np.concatenate([f[:, 1], f[:, 2]]))), shape=(n, n))
    _, label = connected_components(g, directed=False)
    face_label = label[f[:, 0]]
    cells = np.bincount(face_label, minlength=label.max() + 1)
    keep_face = cells[face_label] > min_cells
    f = f[keep_face]
    used = np.zeros(n, dtype=bool)
    used[f.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return verts[used], remap[f].astype(np.int32)


def smooth_mesh(input_mesh: Mesh, num_iterations: int = 150, relaxation_factor: float = 0.01) -> Mesh:
    """vtkSmoothPolyDataFilter with its defaults (relaxation 0.01, boundary smoothing on, no feature edges)."""
    lib = _lib.load()
    n = len(input_mesh.verts)
    if n == 0 or num_iterations <= 0:
        return Mesh(input_mesh.verts.copy(), input_mesh.faces.copy(), dict(input_mesh.point_data))
    off, nbr = vertex_adjacency(n, input_mesh.faces)
    v_in, d_off, d_nbr = _dev(input_mesh.verts, np.float32), _dev(off, np.int32), _dev(nbr, np.int32)
    tmp, out = torch.empty_like(v_in), torch.empty_like(v_in)
    with torch.cuda.device(v_in.device):
        _lib.check(lib.oai_mesh_smooth(v_in.data_ptr(), n, d_off.data_ptr(), d_nbr.data_ptr(), int(num_iterations), float(relaxation_factor),
                                       tmp.data_ptr(), out.data_ptr(), _stream()), "oai_mesh_smooth")
    return Mesh(out.cpu().numpy(), input_mesh.faces.copy(), dict(input_mesh.point_data))


def get_mesh(itk_image, num_iterations: int = 150, min_cells: int = 3000) -> Mesh:
    """mesh_processing.py:325-340: iso-surface of the probability map at 0.5 in (x,y,z)*spacing, small regions dropped
    (get_vtk_mesh), then smoothed."""
    img = as_image(itk_image)
    verts, faces = marching_cubes(np.asarray(img.array, dtype=np.float32), 0.5, img.spacing)
    verts, faces = keep_large_regions(verts, faces, min_cells)
    return smooth_mesh(Mesh(verts, faces), num_iterations=num_iterations)


# ---- per-cell attributes (trimesh in the reference) ------------------------------------------------------------------------------
def get_cell_centroid(mesh: Mesh) -> np.ndarray:
    v = mesh.verts.astype(np.float64)
    return v[mesh.faces].sum(axis=1) / 3.0


def get_cell_normals(mesh: Mesh) -> np.ndarray:
    v = mesh.verts.astype(np.float64)
    a, b, c = v[mesh.faces[:, 0]], v[mesh.faces[:, 1]], v[mesh.faces[:, 2]]
    n = np.cross(b - a, c - a)
    length = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.where(length > 0, length, 1.0)


def get_sub_mesh(mesh: Mesh, face_list: np.ndarray) -> Mesh:
    """get_vtk_sub_mesh (:150-194): the selected faces with points renumbered in order of first use."""
    f = mesh.faces[np.asarray(face_list, dtype=np.int64)]
    flat = f.reshape(-1)
    uniq, first = np.unique(flat, return_index=True)
    order = uniq[np.argsort(first)]
    remap = np.full(len(mesh.verts), -1, dtype=np.int64)
    remap[order] = np.arange(len(order))
    return Mesh(mesh.verts[order], remap[f].astype(np.int32))


def split_tibial_cartilage_surface(mesh: Mesh, mesh_normals, mesh_centroids):
    """mesh_processing.py:197-223"""
    from sklearn.cluster import KMeans
    cn = (mesh_centroids - np.mean(mesh_centroids, axis=0)) / (np.max(mesh_centroids, axis=0) - np.min(mesh_centroids, axis=0))
    features = np.concatenate((cn * 1, mesh_normals * 10), axis=1)
    labels = KMeans(n_clusters=2, algorithm="lloyd", random_state=5).fit(features).labels_
    io = labels * 2 - 1
    if mesh_normals[io == -1, 1].mean() < 0:
        io = -io
    inner, outer = np.where(io == -1)[0], np.where(io == 1)[0]
    return get_sub_mesh(mesh, inner), get_sub_mesh(mesh, outer), inner, outer


def cluster_and_segment(mesh_centroids_normalized, face_normal_value, dot_output):
    """mesh_processing.py:227-240"""
    from sklearn.cluster import KMeans
    features = np.concatenate((mesh_centroids_normalized * 1, face_normal_value, dot_output), axis=1)
    labels = KMeans(n_clusters=2, algorithm="lloyd", n_init=5, random_state=5).fit(features).labels_ * 2 - 1
    if face_normal_value[labels == -1, 1].mean() < 0:
        labels = -labels
    return labels


def split_femoral_cartilage_surface(mesh: Mesh, face_normal, face_centroid, num_divisions: int = 3):
    """mesh_processing.py:243-294: KMeans per x-slab on (centroid, normal, (bbox centre - centroid) * normal)"""
    cn = (face_centroid - np.mean(face_centroid, axis=0)) / (np.max(face_centroid, axis=0) - np.min(face_centroid, axis=0))
    xmin, xmax, ymin, ymax, zmin, zmax = mesh.GetBounds()
    center = (np.array([xmin, ymin, zmin]) + np.array([xmax, ymax, zmax])) / 2
    dot_output = np.multiply(center - face_centroid, face_normal)
    x_coord = cn[:, 0]
    io = np.zeros(cn.shape[0])
    min_x, max_x = np.min(x_coord), np.max(x_coord)
    step = (max_x - min_x) / num_divisions
    for i in range(num_divisions):
        lower = min_x + step * i
        idx = np.where((x_coord >= lower) & (x_coord < lower + step))[0]
        if len(idx) < 2:
            continue
        np.put(io, idx, cluster_and_segment(cn[idx], face_normal[idx], dot_output[idx]))
    inner, outer = np.where(io == -1)[0], np.where(io == 1)[0]
    return get_sub_mesh(mesh, inner), get_sub_mesh(mesh, outer), inner, outer


def split_mesh(mesh: Mesh, mesh_type: str = "FC") -> Tuple[Mesh, Mesh]:
    """mesh_processing.py:353-378"""
    normals, centroids = get_cell_normals(mesh), get_cell_centroid(mesh)
    if mesh_type == "FC":
        inner, outer, _, _ = split_femoral_cartilage_surface(mesh, normals, centroids)
    else:
        inner, outer, _, _ = split_tibial_cartilage_surface(mesh, normals, centroids)
    return inner, outer


# ---- thickness -----------------------------------------------------------------------------------------------------------------
def point_distance(points: np.ndarray, mesh: Mesh, broad_phase: bool = True) -> np.ndarray:
    """Unsigned distance from each point to the mesh surface.  ``broad_phase``: bin the triangles into a uniform grid whose cell is
    the longest triangle edge (>= 2 voxels' worth) so that a point only tests the triangles around it; False = brute force."""
    lib = _lib.load()
    p, v, f = _dev(points, np.float32), _dev(mesh.verts, np.float32), _dev(mesh.faces, np.int32)
    out = torch.empty(len(points), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        if broad_phase and len(mesh.faces) > 0:
            tri = mesh.verts[mesh.faces].astype(np.float64)
            edge = max(np.linalg.norm(tri[:, 0] - tri[:, 1], axis=1).max(), np.linalg.norm(tri[:, 1] - tri[:, 2], axis=1).max(),
                       np.linalg.norm(tri[:, 2] - tri[:, 0], axis=1).max())
            lo, hi = mesh.verts.min(axis=0).astype(np.float64), mesh.verts.max(axis=0).astype(np.float64)
            h = max(float(edge) * 1.0001, float((hi - lo).max()) / 512.0, 1e-6)              # at most 512 cells per axis
            dims = np.maximum(np.ceil((hi - lo) / h).astype(np.int64) + 1, 1)
            glo = (C.c_float * 3)(*[float(x) for x in lo - 0.5 * h * 1e-3])
            gd = (C.c_int * 3)(*[int(x) for x in dims])
            ws = torch.empty(int(lib.oai_mesh_grid_workspace_bytes(gd, len(mesh.faces))), dtype=torch.uint8, device=p.device)
            _lib.check(lib.oai_mesh_point_distance_grid(p.data_ptr(), len(points), v.data_ptr(), f.data_ptr(), len(mesh.faces), glo, float(h), gd,
                                                        ws.data_ptr(), ws.numel(), out.data_ptr(), _stream()), "oai_mesh_point_distance_grid")
        else:
            _lib.check(lib.oai_mesh_point_distance(p.data_ptr(), len(points), v.data_ptr(), f.data_ptr(), len(mesh.faces), out.data_ptr(), _stream()),
                       "oai_mesh_point_distance")
    return out.cpu().numpy()


def get_distance(inner_mesh: Mesh, outer_mesh: Mesh) -> Tuple[Mesh, Mesh]:
    """vtkDistancePolyDataFilter (:310-322): every point of each mesh gets the unsigned distance to the other mesh's surface
    as point data "Distance"."""
    d_in = point_distance(inner_mesh.verts, outer_mesh)
    d_out = point_distance(outer_mesh.verts, inner_mesh)
    return (Mesh(inner_mesh.verts, inner_mesh.faces, {**inner_mesh.point_data, "Distance": d_in}),
            Mesh(outer_mesh.verts, outer_mesh.faces, {**outer_mesh.point_data, "Distance": d_out}))


def get_thickness_mesh(itk_image, mesh_type: str = "FC", num_iterations: int = 150, min_cells: int = 3000) -> Tuple[Mesh, Mesh]:
    """mesh_processing.py:381-395 (which, like this, always smooths with 150 iterations)."""
    mesh = get_mesh(itk_image, num_iterations=150, min_cells=min_cells)
    inner, outer = split_mesh(mesh, mesh_type)
    return get_distance(inner, outer)
