"""Array-bundle stand-in for the reference's mesh objects.

The reference reads meshes through pyexodus / h5py / salvus (``multi_mesh/io/exodus.py:9-142``,
``components/salvus_mesh_reader.py``), none of which exist in this image; mesh I/O stays host
Python and is out of the hot path's scope.  :class:`HexMesh` carries exactly the
attributes the hot path reads from those objects -- ``points``, ``connectivity`` (0-based, exodus
hex8 corner order), ``nelem``, ``npoint``, ``nodes_per_element``, ``ndim``, named nodal fields --
with the same method names (``get_element_centroid``, ``get_nodal_field``, ``attach_field``), so
the drivers in :mod:`multimesh_amd.api` read like the reference's.
"""
from __future__ import annotations

import numpy as np


class HexMesh:
    """Hexahedral (hex8) mesh held as flat arrays."""

    def __init__(self, points, connectivity, nodal_fields=None):
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        self.connectivity = np.ascontiguousarray(connectivity, dtype=np.int64)
        if self.points.ndim != 2 or self.connectivity.ndim != 2:
            raise ValueError("points must be [npoint, ndim] and connectivity [nelem, nodes_per_element]")
        self.nodal_fields = {}
        for name, values in (nodal_fields or {}).items():
            self.attach_field(name, values)

    # ---- the attributes reference io/exodus.py:29-47 exposes ----
    @property
    def ndim(self):
        return self.points.shape[1]

    @property
    def npoint(self):
        return self.points.shape[0]

    @property
    def nelem(self):
        return self.connectivity.shape[0]

    @property
    def nodes_per_element(self):
        return self.connectivity.shape[1]

    @property
    def nodal_parameters(self):
        return list(self.nodal_fields)

    # ---- methods ----
    def get_element_centroid(self):
        """Element centroids on the GPU (reference io/exodus.py:49-64 -> src/centroid.c)."""
        from .helpers import check, load_lib

        lib = load_lib()
        centroid = np.zeros((self.nelem, self.ndim))
        lib.centroid(self.ndim, self.nelem, self.nodes_per_element, self.connectivity, self.points, centroid)
        check(lib.mm_last_status(), "centroid")
        return centroid

    def get_nodal_field(self, name):
        """reference io/exodus.py:99-111"""
        if name not in self.nodal_fields:
            raise ValueError(f"Could not find the requested field: {name}")
        return self.nodal_fields[name]

    def attach_field(self, name, values):
        """Nodal field of length npoint (reference io/exodus.py:66-97 for the nodal case)."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        if values.shape != (self.npoint,):
            raise ValueError(f"field {name!r} must have shape ({self.npoint},)")
        self.nodal_fields[name] = values

    def fields_matrix(self, names):
        """f64[C, npoint]: one contiguous row per parameter, the layout the gather kernel reads."""
        return np.ascontiguousarray(np.stack([self.get_nodal_field(n) for n in names]))
