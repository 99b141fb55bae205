"""Thin ctypes layer over the device-pointer API of ``multi_mesh_hip.so``.

Arrays handed to a :class:`Context` method may be NumPy arrays (copied to HBM for the call),
:class:`DeviceArray` objects, or anything exposing ``data_ptr()`` / ``shape`` / ``dtype``
(``torch`` CUDA tensors -- torch is only plumbing for device memory and RCCL here).  Results
are :class:`DeviceArray` objects; ``.numpy()`` copies them back.

No CPU fallback: every method ends in a HIP kernel launch and raises
:class:`multimesh_amd.helpers.MultiMeshHipError` when the GPU is missing.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .helpers import MM_FP_EXACT, MM_FP_TOL, MM_KNN_MAX_K, STAGES, MultiMeshHipError, check, load_lib

_NP2ITEM = {np.dtype(np.float64): 8, np.dtype(np.int64): 8, np.dtype(np.int32): 4, np.dtype(np.uint8): 1}


class DeviceArray:
    """A C-contiguous array resident in HBM, owned (or merely viewed) by a Context."""

    def __init__(self, ctx, ptr, shape, dtype, owner=True, keepalive=None):
        self.ctx = ctx
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner
        self._keepalive = keepalive

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def data_ptr(self):
        return self.ptr

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            check(self.ctx.lib.mm_copy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes), "mm_copy_d2h")
        return out

    def rows(self, start, stop):
        """A non-owning view of rows [start, stop) (first axis)."""
        row_bytes = self.nbytes // max(self.shape[0], 1) if self.shape[0] else 0
        return DeviceArray(self.ctx, self.ptr + start * row_bytes, (stop - start,) + self.shape[1:], self.dtype,
                           owner=False, keepalive=self)

    def free(self):
        if self._owner and self.ptr and self.ctx.handle:
            self.ctx.lib.mm_device_free(self.ctx.handle, self.ptr)
        self.ptr = 0
        self._owner = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class KnnIndex:
    """Device-resident search structure over source points (the cKDTree stand-in)."""

    def __init__(self, ctx, handle, nsrc, ndim, keepalive):
        self.ctx, self.handle, self.nsrc, self.ndim = ctx, handle, nsrc, ndim
        self._keepalive = keepalive

    def query(self, points, k, want_dist=False):
        """``tree.query(points, k)`` of reference scripts/cli.py:71-73 -> idx int64[N,k] (, dist)."""
        ctx = self.ctx
        pts = ctx.asdevice(points, np.float64)
        if len(pts.shape) != 2 or pts.shape[1] != self.ndim:
            raise ValueError("points must be [N, ndim]")
        if not 0 <= k <= MM_KNN_MAX_K:
            raise ValueError(f"k must be in 0..{MM_KNN_MAX_K}")
        n = pts.shape[0]
        idx = ctx.empty((n, k), np.int64)
        dist = ctx.empty((n, k), np.float64) if want_dist else None
        check(ctx.lib.mm_knn_query(ctx.handle, self.handle, pts.ptr, n, k, idx.ptr, dist.ptr if dist else None),
              "mm_knn_query")
        return (idx, dist) if want_dist else idx

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.mm_knn_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Source:
    """A hex8 source mesh kept resident for repeated calls (mm_source_create): nodes, connectivity, element centroids and
    the search grid over them -- built once, like the reference's cKDTree (scripts/cli.py:66, queried at :141-195)."""

    def __init__(self, ctx, handle, nodes, conn):
        self.ctx, self.handle = ctx, handle
        self.nodes, self.conn = nodes, conn      # (borrowed by the library: kept alive here)

    def interpolate(self, points, fields, nelem_to_search=20, want_operator=False, out=None):
        """:meth:`Context.interpolate_hex8` without the centroid and grid-build stages; identical results."""
        ctx = self.ctx
        pts = ctx.asdevice(points, np.float64)
        f = ctx.asdevice(fields, np.float64)
        if len(f.shape) == 1:
            f = DeviceArray(ctx, f.ptr, (1, f.shape[0]), f.dtype, owner=False, keepalive=f)
        if f.shape[1] != self.nodes.shape[0]:
            raise ValueError("fields must be [C, number of nodes]")
        n, ncomp = pts.shape[0], f.shape[0]
        out = ctx.empty((n, ncomp), np.float64) if out is None else ctx.asdevice(out, np.float64)
        enc = ctx.empty((n, 8), np.int64) if want_operator else None
        w = ctx.empty((n, 8), np.float64) if want_operator else None
        nf = check(ctx.lib.mm_interpolate_hex8_on(ctx.handle, self.handle, pts.ptr, n, f.ptr, ncomp, nelem_to_search, out.ptr,
                                                  enc.ptr if enc else None, w.ptr if w else None), "mm_interpolate_hex8_on")
        if want_operator:
            return out, enc, w, int(nf)
        return out, int(nf)

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.mm_source_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream.  ``stream`` is a raw hipStream_t value (e.g.
    ``torch.cuda.current_stream().cuda_stream``); None = the device's default stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load_lib()
        h = C.c_void_p()
        check(self.lib.mm_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)),
              "mm_context_create")
        self.handle = h.value
        self.device = int(device)

    # ---- memory -------------------------------------------------------------------------
    def empty(self, shape, dtype):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        dtype = np.dtype(dtype)
        p = C.c_void_p()
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        check(self.lib.mm_device_alloc(self.handle, nbytes, C.byref(p)), "mm_device_alloc")
        return DeviceArray(self, p.value, shape, dtype)

    def zeros(self, shape, dtype):
        a = self.empty(shape, dtype)
        if a.nbytes:
            check(self.lib.mm_memset(self.handle, a.ptr, 0, a.nbytes), "mm_memset")
        return a

    def to_device(self, array, dtype=None):
        a = np.ascontiguousarray(array, dtype=dtype)
        d = self.empty(a.shape, a.dtype)
        if a.nbytes:
            check(self.lib.mm_copy_h2d(self.handle, d.ptr, a.ctypes.data, a.nbytes), "mm_copy_h2d")
        return d

    def asdevice(self, x, dtype):
        """NumPy -> copy to HBM; DeviceArray / torch-like -> wrap without copying."""
        dtype = np.dtype(dtype)
        if isinstance(x, DeviceArray):
            if x.dtype != dtype:
                raise TypeError(f"expected {dtype}, got {x.dtype}")
            return x
        if hasattr(x, "data_ptr") and hasattr(x, "shape"):
            name = str(getattr(x, "dtype", "")).replace("torch.", "")
            if name and np.dtype(name) != dtype:
                raise TypeError(f"expected {dtype}, got {name}")
            if hasattr(x, "is_contiguous") and not x.is_contiguous():
                raise ValueError("device tensor must be contiguous")
            return DeviceArray(self, x.data_ptr(), tuple(x.shape), dtype, owner=False, keepalive=x)
        return self.to_device(np.asarray(x), dtype)

    def synchronize(self):
        check(self.lib.mm_synchronize(self.handle), "mm_synchronize")

    # ---- timers -------------------------------------------------------------------------
    def set_profiling(self, on=True):
        """Stage timers: False / True (all stages) / 2 (only the kNN tile kernel and the first locate pass)."""
        check(self.lib.mm_set_profiling(self.handle, 2 if on == 2 and on is not True else (1 if on else 0)), "mm_set_profiling")

    def set_lazy_lists(self, on=True):
        """interpolate_hex8 asks the kNN stage for the 8 nearest first and for the full list only for
        targets that exhaust them (bit-identical outputs; default on)."""
        check(self.lib.mm_set_lazy_lists(self.handle, 1 if on else 0), "mm_set_lazy_lists")

    def set_fp_mode(self, mode):
        """Arithmetic of the hex8 locate stage: "exact" (default: the reference's operations, every output bit-identical)
        or "tol" (cheaper Newton arithmetic that certifies every decision of the reference's iteration or repeats the
        solve exactly: node ids / failed count still bit-identical, weights and values to ~1e-12; include/multimesh_hip.h)."""
        m = {"exact": MM_FP_EXACT, "tol": MM_FP_TOL, MM_FP_EXACT: MM_FP_EXACT, MM_FP_TOL: MM_FP_TOL}[mode]
        check(self.lib.mm_set_fp_mode(self.handle, m), "mm_set_fp_mode")

    def fp_mode(self):
        return "tol" if check(self.lib.mm_get_fp_mode(self.handle), "mm_get_fp_mode") == MM_FP_TOL else "exact"

    def last_locate_stats(self):
        """Of the last hex8 locate stage: solves MM_FP_TOL repeated in the reference's arithmetic, targets that went
        through the reference-order kernel, targets of a long on-demand list's second pass (synchronises)."""
        buf = (C.c_longlong * 4)()
        check(self.lib.mm_last_locate_stats(self.handle, buf), "mm_last_locate_stats")
        return {"redone_exact": int(buf[0]), "reference_order": int(buf[1]), "second_pass": int(buf[2])}

    def last_timings(self):
        """Per-stage milliseconds of the last call (hipEvents on this context's stream)."""
        buf = (C.c_double * len(STAGES))()
        check(self.lib.mm_last_timings(self.handle, buf, len(STAGES)), "mm_last_timings")
        return {name: buf[i] for i, name in enumerate(STAGES)}

    # ---- A1 -----------------------------------------------------------------------------
    def centroid(self, connectivity, points):
        conn = self.asdevice(connectivity, np.int64)
        pts = self.asdevice(points, np.float64)
        nelem, nper = conn.shape
        ndim = pts.shape[1]
        out = self.empty((nelem, ndim), np.float64)
        check(self.lib.mm_centroid(self.handle, ndim, nelem, nper, conn.ptr, pts.ptr, out.ptr), "mm_centroid")
        return out

    # ---- A2 -----------------------------------------------------------------------------
    def knn_build(self, sources):
        src = self.asdevice(sources, np.float64)
        if len(src.shape) != 2 or not 1 <= src.shape[1] <= 3:
            raise ValueError("sources must be [nsrc, ndim] with ndim in 1..3")
        h = C.c_void_p()
        check(self.lib.mm_knn_build(self.handle, src.ptr, src.shape[0], src.shape[1], C.byref(h)), "mm_knn_build")
        return KnnIndex(self, h.value, src.shape[0], src.shape[1], keepalive=src)

    # ---- A4 -----------------------------------------------------------------------------
    def locate_hex8(self, nearest_element_indices, connectivity, nodes, points, enc=None, weights=None,
                    conn_is_exodus=False):
        """Returns (enc int64[N,8], weights f64[N,8], nfailed).  ``enc``/``weights`` given ->
        updated in place (rows of failed points untouched), else zero-initialised here as the
        reference's callers do (scripts/cli.py:77-78)."""
        nn = self.asdevice(nearest_element_indices, np.int64)
        conn = self.asdevice(connectivity, np.int64)
        nod = self.asdevice(nodes, np.float64)
        pts = self.asdevice(points, np.float64)
        n = pts.shape[0]
        k = nn.shape[1] if len(nn.shape) == 2 else 0
        if conn.shape[1] != 8 or nod.shape[1] != 3 or pts.shape[1] != 3 or nn.shape[0] != n:
            raise ValueError("shape mismatch: need nn[N,k], connectivity[E,8], nodes[M,3], points[N,3]")
        enc = self.zeros((n, 8), np.int64) if enc is None else self.asdevice(enc, np.int64)
        w = self.zeros((n, 8), np.float64) if weights is None else self.asdevice(weights, np.float64)
        nf = check(self.lib.mm_locate_hex8(self.handle, k, n, nn.ptr, conn.ptr, conn.shape[0],
                                           1 if conn_is_exodus else 0, enc.ptr, nod.ptr, w.ptr, pts.ptr),
                   "mm_locate_hex8")
        return enc, w, int(nf)

    # ---- A9 -----------------------------------------------------------------------------
    def gather(self, fields, ids, weights, point_major=True):
        """fields f64[C,M] (or [M]) -> f64[N,C] (point_major) or f64[C,N]."""
        f = self.asdevice(fields, np.float64)
        if len(f.shape) == 1:
            f = DeviceArray(self, f.ptr, (1, f.shape[0]), f.dtype, owner=False, keepalive=f)
        idv = self.asdevice(ids, np.int64)
        w = self.asdevice(weights, np.float64)
        if idv.shape != w.shape or len(idv.shape) != 2:
            raise ValueError("ids and weights must both be [N, P]")
        n, p = idv.shape
        ncomp, nsrc = f.shape
        out = self.empty((n, ncomp) if point_major else (ncomp, n), np.float64)
        check(self.lib.mm_gather(self.handle, f.ptr, nsrc, ncomp, idv.ptr, w.ptr, n, p, out.ptr,
                                 1 if point_major else 0), "mm_gather")
        return out

    # ---- A10 (GLL) ------------------------------------------------------------------------
    def locate_gll(self, shape_order, nearest_element_indices, gll_points, points, tolerance=1.05,
                   snap_to_nearest=False):
        """``get_element_weights`` core (reference interpolator.py:1181-1233) for
        gll_points f64[E, (order+1)^dim, dim].  Returns (elem int64[N], coeffs f64[N,P], nmissing)."""
        nn = self.asdevice(nearest_element_indices, np.int64)
        gp = self.asdevice(gll_points, np.float64)
        pts = self.asdevice(points, np.float64)
        nelem, P, dim = gp.shape
        if P != (shape_order + 1) ** dim or pts.shape[1] != dim:
            raise ValueError("gll_points must be [nelem, (order+1)^dim, dim] and points [N, dim]")
        n = pts.shape[0]
        k = nn.shape[1] if len(nn.shape) == 2 else 0
        elem = self.empty((n,), np.int64)
        coeffs = self.empty((n, P), np.float64)
        miss = check(self.lib.mm_locate_gll(self.handle, shape_order, dim, k, n, nn.ptr, gp.ptr, nelem, pts.ptr,
                                            float(tolerance), 1 if snap_to_nearest else 0, elem.ptr, coeffs.ptr),
                     "mm_locate_gll")
        return elem, coeffs, int(miss)

    def locate_gll_bbox(self, shape_order, nearest_element_indices, gll_points, points):
        """The bounding-box variant ``_check_if_inside_element`` (reference interpolator.py:1409-1473)
        for gll_points f64[E, (order+1)^dim, dim].  Returns (elem int64[N], coeffs f64[N,P], number
        of points whose final inverse transform failed)."""
        nn = self.asdevice(nearest_element_indices, np.int64)
        gp = self.asdevice(gll_points, np.float64)
        pts = self.asdevice(points, np.float64)
        nelem, P, dim = gp.shape
        if P != (shape_order + 1) ** dim or pts.shape[1] != dim:
            raise ValueError("gll_points must be [nelem, (order+1)^dim, dim] and points [N, dim]")
        n = pts.shape[0]
        k = nn.shape[1] if len(nn.shape) == 2 else 0
        elem = self.empty((n,), np.int64)
        coeffs = self.empty((n, P), np.float64)
        hard = check(self.lib.mm_locate_gll_bbox(self.handle, shape_order, dim, k, n, nn.ptr, gp.ptr, nelem, pts.ptr,
                                                 elem.ptr, coeffs.ptr), "mm_locate_gll_bbox")
        return elem, coeffs, int(hard)

    def gather_elem(self, element_nodal_fields, elem, coeffs, point_major=True):
        """``np.sum(coeffs * field[elem], axis=1)`` (reference interpolator.py:976);
        element_nodal_fields f64[C, E, P] (or [E, P]) -> f64[N, C]."""
        f = self.asdevice(element_nodal_fields, np.float64)
        if len(f.shape) == 2:
            f = DeviceArray(self, f.ptr, (1,) + f.shape, f.dtype, owner=False, keepalive=f)
        el = self.asdevice(elem, np.int64)
        co = self.asdevice(coeffs, np.float64)
        ncomp, nelem, P = f.shape
        n = el.shape[0]
        if co.shape != (n, P):
            raise ValueError("coeffs must be [N, P]")
        out = self.empty((n, ncomp) if point_major else (ncomp, n), np.float64)
        check(self.lib.mm_gather_elem(self.handle, f.ptr, nelem, ncomp, el.ptr, co.ptr, n, P, out.ptr,
                                      1 if point_major else 0), "mm_gather_elem")
        return out

    # ---- fused ---------------------------------------------------------------------------
    def interpolate_gll(self, shape_order, gll_points, points, element_nodal_fields, nelem_to_search=20,
                        tolerance=1.05, snap_to_nearest=False, want_operator=False, out=None):
        """The GLL form of the whole path (reference interpolator.py:931-977) on resident arrays:
        gll_points f64[E, P, dim], points f64[N, dim], element_nodal_fields f64[C, E, P].
        Returns (values f64[N, C], nmissing) or (values, elem int64[N], coeffs f64[N, P], nmissing)."""
        gp = self.asdevice(gll_points, np.float64)
        pts = self.asdevice(points, np.float64)
        f = self.asdevice(element_nodal_fields, np.float64)
        if len(f.shape) == 2:
            f = DeviceArray(self, f.ptr, (1,) + f.shape, f.dtype, owner=False, keepalive=f)
        nelem, P, dim = gp.shape
        if P != (shape_order + 1) ** dim or len(pts.shape) != 2 or pts.shape[1] != dim:
            raise ValueError("gll_points must be [nelem, (order+1)^dim, dim] and points [N, dim]")
        if f.shape[1:] != (nelem, P):
            raise ValueError("element_nodal_fields must be [C, nelem, P]")
        n, ncomp = pts.shape[0], f.shape[0]
        if out is None:
            out = self.empty((n, ncomp), np.float64)
        else:
            out = self.asdevice(out, np.float64)
            if out.shape != (n, ncomp):
                raise ValueError("out must be [N, C]")
        elem = self.empty((n,), np.int64) if want_operator else None
        coeffs = self.empty((n, P), np.float64) if want_operator else None
        miss = check(self.lib.mm_interpolate_gll(self.handle, shape_order, dim, gp.ptr, nelem, pts.ptr, n, f.ptr, ncomp,
                                                 nelem_to_search, float(tolerance), 1 if snap_to_nearest else 0,
                                                 out.ptr, elem.ptr if elem else None, coeffs.ptr if coeffs else None),
                     "mm_interpolate_gll")
        if want_operator:
            return out, elem, coeffs, int(miss)
        return out, int(miss)

    def interpolate_hex8(self, nodes, connectivity, points, fields, nelem_to_search=20, want_operator=False,
                         out=None):
        """The whole hot path of reference scripts/cli.py:62-100 on resident arrays.
        connectivity is the mesh's own (exodus-order) hex8 connectivity.
        Returns (values f64[N,C], nfailed) or (values, enc, weights, nfailed)."""
        nod = self.asdevice(nodes, np.float64)
        conn = self.asdevice(connectivity, np.int64)
        pts = self.asdevice(points, np.float64)
        f = self.asdevice(fields, np.float64)
        if len(f.shape) == 1:
            f = DeviceArray(self, f.ptr, (1, f.shape[0]), f.dtype, owner=False, keepalive=f)
        n = pts.shape[0]
        ncomp = f.shape[0]
        if f.shape[1] != nod.shape[0]:
            raise ValueError("fields must be [C, number of nodes]")
        out = self.empty((n, ncomp), np.float64) if out is None else self.asdevice(out, np.float64)
        enc = self.empty((n, 8), np.int64) if want_operator else None
        w = self.empty((n, 8), np.float64) if want_operator else None
        nf = check(self.lib.mm_interpolate_hex8(self.handle, nod.ptr, nod.shape[0], conn.ptr, conn.shape[0],
                                                pts.ptr, n, f.ptr, ncomp, nelem_to_search, out.ptr,
                                                enc.ptr if enc else None, w.ptr if w else None),
                   "mm_interpolate_hex8")
        if want_operator:
            return out, enc, w, int(nf)
        return out, int(nf)

    def source(self, nodes, connectivity):
        """Keep a hex8 source mesh resident (centroids + search grid built once): :class:`Source`."""
        nod = self.asdevice(nodes, np.float64)
        conn = self.asdevice(connectivity, np.int64)
        h = C.c_void_p()
        check(self.lib.mm_source_create(self.handle, nod.ptr, nod.shape[0], conn.ptr, conn.shape[0], C.byref(h)), "mm_source_create")
        return Source(self, h.value, nod, conn)

    def interpolate_hex8_host(self, nodes, connectivity, points, fields, nelem_to_search=20, want_operator=False,
                              out=None):
        """:meth:`interpolate_hex8` for NumPy arrays on the host (reference scripts/cli.py:62-100 holds
        nothing else): uploads overlapped with the kernels, device copies cached in the context.
        Returns NumPy arrays: (values f64[N,C], nfailed) or (values, enc, weights, nfailed)."""
        nod = np.ascontiguousarray(nodes, dtype=np.float64)
        conn = np.ascontiguousarray(connectivity, dtype=np.int64)
        pts = np.ascontiguousarray(points, dtype=np.float64)
        f = np.ascontiguousarray(fields, dtype=np.float64)
        if f.ndim == 1:
            f = f[None]
        if nod.ndim != 2 or nod.shape[1] != 3 or conn.ndim != 2 or conn.shape[1] != 8 or pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("need nodes[M,3], connectivity[E,8], points[N,3]")
        if f.shape[1] != nod.shape[0]:
            raise ValueError("fields must be [C, number of nodes]")
        n, ncomp = pts.shape[0], f.shape[0]
        if out is None:
            out = np.empty((n, ncomp), dtype=np.float64)
        elif out.shape != (n, ncomp) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous f64[N, C] array")
        enc = np.empty((n, 8), dtype=np.int64) if want_operator else None
        w = np.empty((n, 8), dtype=np.float64) if want_operator else None
        nf = check(self.lib.mm_interpolate_hex8_host(self.handle, nod.ctypes.data, nod.shape[0], conn.ctypes.data,
                                                     conn.shape[0], pts.ctypes.data, n, f.ctypes.data, ncomp,
                                                     nelem_to_search, out.ctypes.data,
                                                     enc.ctypes.data if want_operator else None,
                                                     w.ctypes.data if want_operator else None),
                   "mm_interpolate_hex8_host")
        if want_operator:
            return out, enc, w, int(nf)
        return out, int(nf)

    # ---- section 8f-4: device passes of the layer-aware drivers ------------------------------------
    def scatter_elements(self, values, inverse, elem_ids, out):
        """``out[:, elem_ids] = values[inverse].reshape(len(elem_ids), P, C)`` transposed to the element-nodal
        layout (reference interpolator.py:1079-1081): values f64[U, C], inverse int64[len(elem_ids) * P],
        out f64[C, E, P] (updated in place on the device)."""
        v = self.asdevice(values, np.float64)
        inv = self.asdevice(inverse, np.int64)
        ids = self.asdevice(elem_ids, np.int64)
        o = self.asdevice(out, np.float64)
        ncomp, nelem_out, P = o.shape
        if v.shape[1] != ncomp or inv.size != ids.size * P:
            raise ValueError("need values[U, C], inverse[len(elem_ids) * P], out[C, E, P]")
        check(self.lib.mm_scatter_elements(self.handle, v.ptr, v.shape[0], ncomp, inv.ptr, ids.ptr, ids.size, P,
                                           nelem_out, o.ptr), "mm_scatter_elements")
        return o

    def fluid_solid_fix(self, values, previous, solid, vs_index):
        """The fix-up of reference interpolator.py:829-841 on values f64[E, C, P] (in place on the device):
        fluid elements and solid elements with a zero shear velocity get ``previous`` back.  Returns the
        number of solid elements restored."""
        v = self.asdevice(values, np.float64)
        prev = self.asdevice(previous, np.float64)
        sol = self.asdevice(np.ascontiguousarray(solid, dtype=np.uint8), np.uint8)
        nelem, ncomp, P = v.shape
        if prev.shape != v.shape or sol.size != nelem:
            raise ValueError("need values[E, C, P], previous[E, C, P], solid[E]")
        return int(check(self.lib.mm_fluid_solid_fix(self.handle, v.ptr, prev.ptr, sol.ptr, nelem, ncomp, P, int(vs_index)),
                         "mm_fluid_solid_fix"))

    # ---- A11 ----------------------------------------------------------------------------
    def unique_points(self, points, unique_out=None, inverse_out=None, ordered=True):
        """``np.unique(points, axis=0, return_inverse=True)`` (reference utils.py:484-488) on the
        device: (unique f64[U, dim] in lexicographic order, inverse int64[N]).  ``unique_out`` f64[N, dim] /
        ``inverse_out`` int64[N]: caller-owned device buffers to write into (no allocation in the call).
        ``ordered=False``: the unique rows in the order of their first occurrence instead (a hash table, 2-3x
        faster) -- enough wherever the rows are only interpolated and scattered back through the inverse."""
        pts = self.asdevice(points, np.float64)
        n, dim = pts.shape
        uniq = self.empty((max(n, 1), dim), np.float64) if unique_out is None else self.asdevice(unique_out, np.float64)
        inv = self.empty((max(n, 1),), np.int64) if inverse_out is None else self.asdevice(inverse_out, np.int64)
        if uniq.size < n * dim or inv.size < n:
            raise ValueError("unique_out / inverse_out too small: need [N, dim] and [N]")
        fn = self.lib.mm_unique_points if ordered else self.lib.mm_unique_points_any_order
        nu = check(fn(self.handle, pts.ptr, n, dim, uniq.ptr, inv.ptr), "mm_unique_points")
        return (DeviceArray(self, uniq.ptr, (int(nu), dim), np.float64, owner=False, keepalive=uniq),
                DeviceArray(self, inv.ptr, (n,), np.int64, owner=False, keepalive=inv))

    def close(self):
        if self.handle:
            self.lib.mm_context_destroy(self.handle)
        self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=0):
    """Process-wide context per device (created on first use; raises without a GPU)."""
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


__all__ = ["Context", "DeviceArray", "KnnIndex", "default_context", "MultiMeshHipError"]
