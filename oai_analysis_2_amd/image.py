"""A minimal stand-in for ``itk.Image``: voxel array + physical-space metadata.

The reference passes ``itk.Image`` objects across its API (analysis_object.py:43-49);
ITK is not installed on the GPU box, so the numpy core of this package works on
``Image`` and thin adapters convert from/to ``itk`` when it is importable.

Conventions (ITK's): ``array`` is indexed [z, y, x]; ``spacing`` / ``origin`` are (x, y, z);
``direction`` is the 3x3 matrix whose columns are the physical directions of the index
axes (x, y, z).  physical = origin + direction @ (spacing * index_xyz).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

import numpy as np


@dataclass
class Image:
    array: np.ndarray
    spacing: np.ndarray = field(default_factory=lambda: np.ones(3))
    origin: np.ndarray = field(default_factory=lambda: np.zeros(3))
    direction: np.ndarray = field(default_factory=lambda: np.eye(3))

    def __post_init__(self):
        self.spacing = np.asarray(self.spacing, dtype=np.float64).reshape(3)
        self.origin = np.asarray(self.origin, dtype=np.float64).reshape(3)
        self.direction = np.asarray(self.direction, dtype=np.float64).reshape(3, 3)

    # numpy protocol, so np.min(image) / np.asarray(image) behave as they do on itk images
    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.array, dtype=dtype)

    @property
    def shape(self):
        return self.array.shape

    @property
    def size_xyz(self) -> np.ndarray:
        return np.asarray(self.array.shape[::-1], dtype=np.int64)

    def CopyInformation(self, other: "Image") -> None:       # itk.Image.CopyInformation
        self.spacing, self.origin, self.direction = other.spacing.copy(), other.origin.copy(), other.direction.copy()

    def like(self, array: np.ndarray) -> "Image":
        return Image(array, self.spacing.copy(), self.origin.copy(), self.direction.copy())

    def index_to_physical_affine(self):
        """(A, b) with physical = A @ index_xyz + b."""
        return self.direction @ np.diag(self.spacing), self.origin.copy()


def as_image(obj) -> Image:
    """Accept an ``Image``, a numpy array (unit spacing) or an ``itk.Image``."""
    if isinstance(obj, Image):
        return obj
    if isinstance(obj, np.ndarray):
        return Image(obj)
    try:  # pragma: no cover - itk is absent in this environment
        import itk
        arr = itk.GetArrayFromImage(obj)
        return Image(arr, np.asarray(obj.GetSpacing()), np.asarray(obj.GetOrigin()),
                     np.asarray(itk.array_from_matrix(obj.GetDirection())))
    except ImportError:
        raise TypeError(f"cannot interpret {type(obj)!r} as an image (itk is not installed)")


def to_itk(img: Image):  # pragma: no cover - itk is absent in this environment
    import itk
    out = itk.GetImageFromArray(np.ascontiguousarray(img.array))
    out.SetSpacing([float(v) for v in img.spacing])
    out.SetOrigin([float(v) for v in img.origin])
    out.SetDirection(itk.matrix_from_array(np.ascontiguousarray(img.direction)))
    return out
