"""Mesh file adapters for the file-level drivers of :mod:`multimesh_amd.api`.

Host Python, outside the hot path: they turn files into the flat arrays the HIP kernels read and
write the results back.  Two containers:

* **Salvus HDF5 meshes** (reference ``components/salvus_mesh_reader.py:38-97``, ``utils.py:137-217``):
  ``MODEL/coordinates`` f64[nelem, P, dim], ``MODEL/data`` [nelem, nparam, P] whose parameter names sit in
  the ``DIMENSION_LABELS`` attribute as ``"[ VP | VS | RHO ]"``, ``MODEL/element_data`` [nelem, nfield].
  Read through **h5py** when it is installed.  Every function here also accepts an already open,
  h5py-like object in place of a path -- anything with ``obj[path]`` -> dataset (``[...]``, ``.shape``,
  ``.attrs``), ``in``, ``del`` and ``create_dataset``; :class:`MemoryH5` is such an object held in memory
  (meshes that never touch a file, and the tests of the label / layout logic in an image without h5py).
* **Exodus II meshes** (reference ``io/exodus.py:29-47``, ``:66-142``; the reference goes through pyexodus):
  an Exodus file is a netCDF file with fixed variable names -- ``connect1`` (1-based), ``coordx/y/z``,
  ``name_nod_var`` / ``vals_nod_var<i>``, ``name_elem_var`` / ``vals_elem_var<i>eb1``.  Classic
  (netCDF-3, CDF-1/2) files are read and updated with ``scipy.io.netcdf_file`` (tested here); HDF5-based
  (netCDF-4) files need h5py, which sees the same variables as root datasets.

Nothing here computes: no interpolation, no trees.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np


# ------------------------------------------------------------------------------------ HDF5
def _require_h5py():
    try:
        import h5py
    except ImportError as e:   # pragma: no cover - depends on the image
        raise ImportError("reading or writing an HDF5 mesh FILE needs h5py, which is not installed; pass an open "
                          "h5py-like object (multimesh_amd.io.MemoryH5) or use the array-level functions of "
                          "multimesh_amd.api") from e
    return h5py


class _MemoryDataset:
    """A NumPy array with the handful of h5py.Dataset members the adapters use."""

    def __init__(self, array):
        self._a = np.array(array)
        self.attrs = {}

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)

    def __getitem__(self, key):
        # (an ndarray key -- a boolean mask, a fancy index -- must not be compared with the empty tuple)
        if isinstance(key, tuple) and len(key) == 0:
            return self._a.copy()
        return self._a[key]

    def __setitem__(self, key, value):
        self._a[key] = value

    def __len__(self):
        return len(self._a)

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)


class MemoryH5:
    """In-memory stand-in for an open ``h5py.File`` / group: slash-separated paths, datasets with
    ``attrs``, ``create_dataset``, ``in``, ``del``.  Usable as a context manager (closing is a no-op)."""

    def __init__(self):
        self._items = {}
        self.attrs = {}

    @staticmethod
    def _split(path):
        return [p for p in str(path).split("/") if p]

    def _walk(self, parts, create=False):
        node = self
        for p in parts:
            if p not in node._items:
                if not create:
                    raise KeyError(p)
                node._items[p] = MemoryH5()
            node = node._items[p]
            if not isinstance(node, MemoryH5):
                raise KeyError(f"{p} is a dataset")
        return node

    def __getitem__(self, path):
        parts = self._split(path)
        return self._walk(parts[:-1])._items[parts[-1]]

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __delitem__(self, path):
        parts = self._split(path)
        del self._walk(parts[:-1])._items[parts[-1]]

    def create_group(self, path):
        return self._walk(self._split(path), create=True)

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        parts = self._split(name)
        parent = self._walk(parts[:-1], create=True)
        if parts[-1] in parent._items:
            raise ValueError(f"{name} exists")
        array = np.zeros(shape, dtype=dtype or np.float64) if data is None else np.array(data, dtype=dtype)
        parent._items[parts[-1]] = _MemoryDataset(array)
        return parent._items[parts[-1]]

    def keys(self):
        return self._items.keys()

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


@contextlib.contextmanager
def open_h5(file, mode="r"):
    """``h5py.File(file, mode)`` for a path; an open h5py-like object is passed through and left open."""
    if isinstance(file, (str, os.PathLike)):
        f = _require_h5py().File(file, mode)
        try:
            yield f
        finally:
            f.close()
    else:
        yield file


def parse_dimension_labels(label, drop_grad=False):
    """``"[ VP | VS | RHO ]"`` -> ``["VP", "VS", "RHO"]`` (bytes or str).  The reference slices the
    brackets off by position (``salvus_mesh_reader.py:67-71``: ``[1:-1]`` after removing blanks;
    ``utils.py:213-214``: ``[2:-2]`` before); ``drop_grad`` also removes the substring ``grad`` as
    ``utils.load_hdf5_params_to_memory`` does for gradient files."""
    if isinstance(label, (bytes, np.bytes_)):
        label = label.decode()
    label = str(label).replace(" ", "")
    if label.startswith("["):
        label = label[1:]
    if label.endswith("]"):
        label = label[:-1]
    if drop_grad:
        label = label.replace("grad", "")
    return label.split("|") if label else []


def dimension_labels(dataset, axis=1, drop_grad=False):
    """Parameter names of axis ``axis`` of an HDF5 dataset (its ``DIMENSION_LABELS`` attribute)."""
    labels = dataset.attrs.get("DIMENSION_LABELS")
    if labels is None:
        raise KeyError("dataset has no DIMENSION_LABELS attribute")
    return parse_dimension_labels(labels[axis], drop_grad)


def set_dimension_labels(dataset, parameters):
    """Label the axes of a ``MODEL/data``-shaped dataset ``element``, ``[ P1 | P2 ]``, ``point`` (reference
    utils.py:159-168, through h5py's dimension-scale API when the dataset has one)."""
    dimstr = "[ " + " | ".join(parameters) + " ]"
    if hasattr(dataset, "dims"):
        dataset.dims[0].label = "element"
        dataset.dims[1].label = dimstr
        dataset.dims[2].label = "point"
    else:
        dataset.attrs["DIMENSION_LABELS"] = np.array([b"element", dimstr.encode(), b"point"])


def pick_parameters(parameters):
    """"TTI" / "ISO" presets or the list itself (reference utils.py:171-189)."""
    if isinstance(parameters, str):
        if parameters == "TTI":
            return ["VPV", "VPH", "VSV", "VSH", "RHO", "ETA", "QKAPPA", "QMU"]
        if parameters == "ISO":
            return ["QKAPPA", "QMU", "RHO", "VP", "VS"]
        return [parameters]
    return list(parameters)


def load_hdf5_params_to_memory(gll, model="MODEL/data", coordinates="MODEL/coordinates"):
    """``(points f64[E, P, dim], data [E, nparam, P], parameter names)`` of a GLL model file (reference
    utils.py:206-217)."""
    with open_h5(gll, "r") as mesh:
        points = np.array(mesh[coordinates][:], dtype=np.float64)
        data = np.array(mesh[model][:])
        params = dimension_labels(mesh[model], 1, drop_grad=True)
    return points, data, params


def remove_and_create_empty_dataset(gll_model, parameters, model="MODEL/data", coordinates="MODEL/coordinates"):
    """Replace ``model`` by a zero f64[nelem, len(parameters), P] dataset labelled with ``parameters`` (reference
    utils.py:137-168).  ``gll_model``: an OPEN writable file."""
    if model in gll_model:
        del gll_model[model]
    shape = gll_model[coordinates].shape
    ds = gll_model.create_dataset(name=model, shape=(shape[0], len(parameters), shape[1]), dtype=np.float64)
    set_dimension_labels(ds, list(parameters))
    return ds


class SalvusMesh:
    """The members of the reference's fast Salvus-mesh reader that its drivers use
    (``components/salvus_mesh_reader.py:8-215``): ``points`` f64[nelem, P, dim], ``nelem``, ``n_gll_points``,
    ``dimensions``, ``shape_order``, ``nodal_parameter_indices`` / ``elemental_parameter_indices`` (the names),
    ``element_nodal_fields`` / ``elemental_fields`` (name -> array, unless ``fast_mode``), the getters and
    ``attach_field`` for EXISTING fields.  ``filename``: a path (h5py) or an open h5py-like object."""

    def __init__(self, filename, fast_mode=True):
        self.filename = filename
        with open_h5(filename, "r") as ds:
            self.points = np.array(ds["MODEL/coordinates"][()], dtype=np.float64)
            self.nodal_parameter_indices = dimension_labels(ds["MODEL/data"], 1)
            self.elemental_parameter_indices = (dimension_labels(ds["MODEL/element_data"], 1)
                                                if "MODEL/element_data" in ds else [])
            self.global_strings = {k: v for k, v in ds["MODEL"].attrs.items() if isinstance(v, np.bytes_)}
            if not fast_mode:
                data = np.array(ds["MODEL/data"][()])
                self.element_nodal_fields = {p: np.ascontiguousarray(data[:, i, :])
                                             for i, p in enumerate(self.nodal_parameter_indices)}
                edata = np.array(ds["MODEL/element_data"][()]) if self.elemental_parameter_indices else None
                self.elemental_fields = {p: np.ascontiguousarray(edata[:, i])
                                         for i, p in enumerate(self.elemental_parameter_indices)}
        self.nelem, self.n_gll_points, self.dimensions = self.points.shape
        self.shape_order = int(np.round(self.n_gll_points ** (1 / self.dimensions)) - 1)

    def get_element_nodes(self):
        return self.points

    def get_element_centroids(self):
        return np.mean(self.points, axis=1)

    def get_element_nodal_field(self, param):
        ind = self.nodal_parameter_indices.index(param)
        with open_h5(self.filename, "r") as ds:
            return np.array(ds["MODEL/data"][:, ind, :])

    def get_elemental_field(self, param):
        ind = self.elemental_parameter_indices.index(param)
        with open_h5(self.filename, "r") as ds:
            return np.array(ds["MODEL/element_data"][:, ind])

    def attach_field(self, name, data):
        """Overwrite an existing element-nodal [nelem, P] or elemental [nelem] field in the file."""
        data = np.asarray(data)
        with open_h5(self.filename, "r+") as ds:
            if data.shape == (self.nelem, self.n_gll_points):
                if name not in self.nodal_parameter_indices:
                    raise ValueError("Currently we only attach existing fields")
                ds["MODEL/data"][:, self.nodal_parameter_indices.index(name), :] = data
                if hasattr(self, "element_nodal_fields"):
                    self.element_nodal_fields[name] = np.array(data, dtype=np.float64)
            elif data.shape == (self.nelem,):
                if name not in self.elemental_parameter_indices:
                    raise ValueError("Currently we only attach existing fields")
                ds["MODEL/element_data"][:, self.elemental_parameter_indices.index(name)] = data
                if hasattr(self, "elemental_fields"):
                    self.elemental_fields[name] = np.array(data)
            else:
                raise ValueError("We can only attach elemental_nodal_field or elemental_fields")


# ------------------------------------------------------------------------------------ Exodus II
def _names(char_rows):
    """Rows of a netCDF char table -> stripped strings."""
    out = []
    for row in np.asarray(char_rows):
        raw = row.tobytes() if hasattr(row, "tobytes") else bytes(row)
        out.append(raw.split(b"\x00", 1)[0].decode("ascii", "replace").strip())
    return out


def _is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == b"\x89HDF\r\n\x1a\n"


class _ExodusContainer:
    """Variables of an Exodus file by name, classic netCDF through scipy or netCDF-4 through h5py."""

    def __init__(self, path, mode):
        self.hdf5 = _is_hdf5(path)
        if self.hdf5:
            self.f = _require_h5py().File(path, "r" if mode == "r" else "r+")
        else:
            from scipy.io import netcdf_file

            self.f = netcdf_file(path, mode, mmap=False)

    def has(self, name):
        return name in (self.f if self.hdf5 else self.f.variables)

    def var(self, name):
        return self.f[name] if self.hdf5 else self.f.variables[name]

    def read(self, name):
        v = self.var(name)
        return np.array(v[()] if self.hdf5 else v[:])

    def write_step(self, name, step, values):
        """``var[step, :] = values`` (time is the record / unlimited axis; grown when the file holds no step yet)"""
        v = self.var(name)
        if self.hdf5 and v.shape[0] <= step:
            v.resize(step + 1, axis=0)   # pragma: no cover - needs h5py
        v[step, :] = values

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.f.close()
        return False


class Exodus:
    """Read / update a single-block Exodus II mesh: the members of the reference's helper (``io/exodus.py``):
    ``ndim``, ``connectivity`` int64[nelem, nodes_per_element] **0-based** (the file is 1-based, :41-43), ``nelem``,
    ``nodes_per_element``, ``points`` f64[npoint, ndim], ``nodal_parameters`` / ``elem_var_names`` (names),
    ``npoint``; ``get_nodal_field`` / ``get_element_field`` (time step 1), ``attach_field`` (mode "a", existing
    variable names only, like the reference's nodal branch), ``get_element_centroid`` (the HIP library)."""

    def __init__(self, filename, mode="r"):
        assert mode in ["a", "r"], "Only mode 'a', 'r' is supported"
        self._filename = os.fspath(filename)
        self.mode = mode
        with _ExodusContainer(self._filename, "r") as e:
            self.connectivity = np.ascontiguousarray(e.read("connect1"), dtype=np.int64) - 1
            if e.has("coord"):
                self.points = np.ascontiguousarray(e.read("coord").T, dtype=np.float64)
            else:
                axes = [e.read(n) for n in ("coordx", "coordy", "coordz") if e.has(n)]
                self.points = np.ascontiguousarray(np.array(axes).T, dtype=np.float64)
            self.nodal_parameters = _names(e.read("name_nod_var")) if e.has("name_nod_var") else []
            self.elem_var_names = _names(e.read("name_elem_var")) if e.has("name_elem_var") else []
        self.nelem, self.nodes_per_element = self.connectivity.shape
        self.ndim = self.points.shape[1]

    @property
    def npoint(self):
        return self.points.shape[0]

    def get_element_centroid(self):
        from .helpers import check, load_lib

        lib = load_lib()
        centroid = np.zeros((self.nelem, self.ndim))
        lib.centroid(self.ndim, self.nelem, self.nodes_per_element, self.connectivity, self.points, centroid)
        check(lib.mm_last_status(), "centroid")
        return centroid

    def _nodal_location(self, e, name):
        """(variable, leading index) of nodal variable ``name`` at a time step: ``vals_nod_var<i>[step, :]`` or the
        older single ``vals_nod_var[step, i-1, :]``"""
        i = self.nodal_parameters.index(name)
        if e.has(f"vals_nod_var{i + 1}"):
            return f"vals_nod_var{i + 1}", None
        return "vals_nod_var", i

    def get_nodal_field(self, name):
        assert name in self.nodal_parameters, "Could not find the requested field"
        with _ExodusContainer(self._filename, "r") as e:
            var, sub = self._nodal_location(e, name)
            data = e.read(var)
            return np.array(data[0] if sub is None else data[0, sub], dtype=np.float64)

    def get_element_field(self, name):
        assert name in self.elem_var_names, "Could not find the requested field"
        with _ExodusContainer(self._filename, "r") as e:
            return np.array(e.read(f"vals_elem_var{self.elem_var_names.index(name) + 1}eb1")[0], dtype=np.float64)

    def attach_field(self, name, values):
        assert self.mode in ["a"], "Attach field option only available in mode 'a'"
        values = np.asarray(values, dtype=np.float64)
        with _ExodusContainer(self._filename, "a") as e:
            if values.size == self.nelem and name in self.elem_var_names:
                e.write_step(f"vals_elem_var{self.elem_var_names.index(name) + 1}eb1", 0, values.ravel())
            elif values.size == self.npoint:
                if name not in self.nodal_parameters:
                    raise ValueError(f"{name!r} is not a nodal variable of {self._filename}")
                var, sub = self._nodal_location(e, name)
                if sub is None:
                    e.write_step(var, 0, values.ravel())
                else:
                    e.var(var)[0, sub, :] = values.ravel()
            else:
                raise ValueError("Shape matches neither the nodes nor the elements")


def write_exodus_classic(filename, points, connectivity, nodal_fields=None, element_fields=None, title="multimesh_amd"):
    """A minimal single-block, one-time-step Exodus II file in classic netCDF (what :class:`Exodus` reads):
    meshes for tests and for handing results to tools that read Exodus.  ``connectivity`` 0-based."""
    from scipy.io import netcdf_file

    points = np.asarray(points, dtype=np.float64)
    connectivity = np.asarray(connectivity)
    nodal_fields, element_fields = dict(nodal_fields or {}), dict(element_fields or {})
    npoint, ndim = points.shape
    nelem, nper = connectivity.shape
    with netcdf_file(filename, "w", version=2) as f:
        f.title = title
        for name, size in (("time_step", None), ("len_string", 33), ("len_name", 33), ("four", 4), ("num_dim", ndim),
                           ("num_nodes", npoint), ("num_elem", nelem), ("num_el_blk", 1),
                           ("num_el_in_blk1", nelem), ("num_nod_per_el1", nper)):
            f.createDimension(name, size)
        f.createVariable("time_whole", "d", ("time_step",))
        for a, axis in enumerate("xyz"[:ndim]):
            f.createVariable("coord" + axis, "d", ("num_nodes",))[:] = points[:, a]
        conn = f.createVariable("connect1", "i", ("num_el_in_blk1", "num_nod_per_el1"))
        conn[:] = connectivity + 1
        conn.elem_type = {8: "HEX8", 4: "QUAD4" if ndim == 2 else "TETRA4", 27: "HEX27"}.get(nper, "UNKNOWN")

        def table(var, dim, names):
            f.createDimension(dim, len(names))
            rows = np.zeros((len(names), 33), dtype="S1")
            for i, n in enumerate(names):
                rows[i, :len(n)] = list(n.encode("ascii").decode("ascii"))
            f.createVariable(var, "c", (dim, "len_name"))[:] = rows

        if nodal_fields:
            table("name_nod_var", "num_nod_var", list(nodal_fields))
        if element_fields:
            table("name_elem_var", "num_elem_var", list(element_fields))
        nod = [f.createVariable(f"vals_nod_var{i + 1}", "d", ("time_step", "num_nodes")) for i in range(len(nodal_fields))]
        ele = [f.createVariable(f"vals_elem_var{i + 1}eb1", "d", ("time_step", "num_el_in_blk1"))
               for i in range(len(element_fields))]
        f.variables["time_whole"][0] = 0.0
        for v, values in zip(nod, nodal_fields.values()):
            v[0, :] = np.asarray(values, dtype=np.float64)
        for v, values in zip(ele, element_fields.values()):
            v[0, :] = np.asarray(values, dtype=np.float64)
