"""NumPy-facing loader for the CPU oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by anything under multimesh_amd/.

Two libraries:

* ``oracle/_build/libmm_oracle.so`` -- our plain-C restatement (mm_oracle.c) of
  the reference hot path (centroid, hex8 locate, NumPy-order gather, brute kNN).
* ``oracle/_ref/multi_mesh_ref.so`` -- the reference's own two C translation
  units compiled where they lie under /root/reference (oracle/Makefile).  It
  pins the restatement and is the CPU baseline ("kind": "reference") in bench.py.
  It exists only if it was built in a container that has /root/reference; the
  prebuilt file travels to the GPU box with the snapshot.

The kNN stage of the reference is ``scipy.spatial.cKDTree`` (scripts/cli.py:66-73),
a third-party dependency that is importable here; ``knn_ckdtree`` calls it exactly
as the reference does and is the kNN oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libmm_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "multi_mesh_ref.so")

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags=["C_CONTIGUOUS"])
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags=["C_CONTIGUOUS"])
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags=["C_CONTIGUOUS"])

_cache = {}


def build(ref: bool = True) -> None:
    """Compile the restatement (and the reference build when its sources are mounted)."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def lib():
    if "o" not in _cache:
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(_HERE, "mm_oracle.c")
        ):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.mmo_centroid.restype = None
        L.mmo_centroid.argtypes = [C.c_int64, C.c_int64, C.c_int64, _i64p, _f64p, _f64p]
        L.mmo_locate_hex8.restype = C.c_int64
        L.mmo_locate_hex8.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _i64p, _f64p,
                                      _f64p, _f64p, C.c_void_p]
        L.mmo_gather.restype = C.c_int
        L.mmo_gather.argtypes = [_f64p, C.c_int64, C.c_int64, _i64p, _f64p, C.c_int64,
                                 C.c_int64, _f64p, C.c_int]
        L.mmo_knn_brute.restype = None
        L.mmo_knn_brute.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_int64,
                                    C.c_int64, _i64p, C.c_void_p]
        L.mmo_hex8_newton.restype = C.c_int
        L.mmo_hex8_newton.argtypes = [_f64p, _f64p, _f64p, C.POINTER(C.c_int)]
        L.mmo_hex8_weights.restype = None
        L.mmo_hex8_weights.argtypes = [_f64p, _f64p]
        L.mmo_gll_coefficients.restype = None
        L.mmo_gll_coefficients.argtypes = [C.c_int, C.c_int, _f64p, _f64p]
        L.mmo_set_gll_strict.restype = None
        L.mmo_set_gll_strict.argtypes = [C.c_int]
        L.mmo_gll_inverse_transform.restype = None
        L.mmo_gll_inverse_transform.argtypes = [C.c_int, C.c_int, _f64p, _f64p, _f64p]
        L.mmo_locate_gll.restype = C.c_int64
        L.mmo_locate_gll.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, _i64p, _f64p, C.c_int64, _f64p,
                                     C.c_double, C.c_int, _i64p, _f64p]
        L.mmo_locate_gll_v1.restype = C.c_int64
        L.mmo_locate_gll_v1.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, _i64p, _f64p, C.c_int64, _f64p, _i64p, _f64p]
        L.mmo_gather_elem.restype = C.c_int
        L.mmo_gather_elem.argtypes = [_f64p, C.c_int64, C.c_int64, _i64p, _f64p, C.c_int64, C.c_int64, _f64p,
                                      C.c_int]
        _cache["o"] = L
    return _cache["o"]


def have_reference() -> bool:
    return os.path.exists(REF_SO)


def reference_lib():
    """The compiled reference (helpers.py:43-81 argtypes, but int64 scalars)."""
    if "r" not in _cache:
        if not have_reference():
            raise FileNotFoundError(REF_SO)
        L = C.CDLL(REF_SO)
        L.centroid.restype = None
        L.centroid.argtypes = [C.c_int64, C.c_int64, C.c_int64, _i64p, _f64p, _f64p]
        L.triLinearInterpolator.restype = C.c_int64
        L.triLinearInterpolator.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _i64p,
                                            _f64p, _f64p, _f64p]
        L.inverseCoordinateTransform.restype = C.c_int
        L.inverseCoordinateTransform.argtypes = [_f64p, _f64p, _f64p]
        L.interpolateAtPoint.restype = None
        L.interpolateAtPoint.argtypes = [_f64p, _f64p]
        _cache["r"] = L
    return _cache["r"]


# ----------------------------------------------------------------------------- restatement
def centroid(conn: np.ndarray, points: np.ndarray) -> np.ndarray:
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    out = np.zeros((conn.shape[0], points.shape[1]))
    lib().mmo_centroid(points.shape[1], conn.shape[0], conn.shape[1], conn, points, out)
    return out


def locate_hex8(nn, conn, nodes, points, want_status=False):
    """Returns (enc int64[N,8], w f64[N,8], nfailed[, status int32[N]])."""
    nn = np.ascontiguousarray(nn, dtype=np.int64)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    n, k = nn.shape if nn.ndim == 2 else (points.shape[0], 0)
    enc = np.zeros((n, 8), dtype=np.int64)
    w = np.zeros((n, 8))
    status = np.zeros(n, dtype=np.int32)
    nf = lib().mmo_locate_hex8(k, n, nn, conn, enc, nodes, w, points,
                               status.ctypes.data_as(C.c_void_p))
    if want_status:
        return enc, w, int(nf), status
    return enc, w, int(nf)


def gather(fields, ids, w, point_major=True):
    """fields f64[C,M] (or [M]) -> f64[N,C] (point_major) or [C,N]."""
    fields = np.ascontiguousarray(np.atleast_2d(fields), dtype=np.float64)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    ncomp, nsrc = fields.shape
    n, p = ids.shape
    out = np.zeros((n, ncomp) if point_major else (ncomp, n))
    rc = lib().mmo_gather(fields, nsrc, ncomp, ids, w, n, p, out, 1 if point_major else 0)
    if rc != 0:
        raise ValueError("mmo_gather: P > 128 unsupported")
    return out


def knn_brute(src, pts, k, want_d2=False):
    src = np.ascontiguousarray(src, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    idx = np.zeros((pts.shape[0], k), dtype=np.int64)
    d2 = np.zeros((pts.shape[0], k))
    lib().mmo_knn_brute(src, src.shape[0], pts, pts.shape[0], src.shape[1], k, idx,
                        d2.ctypes.data_as(C.c_void_p))
    return (idx, d2) if want_d2 else idx


# ----------------------------------------------------------------------------- GLL (parity unpinned)
def gll_coefficients(order, xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    dim = xi.shape[0]
    out = np.zeros((order + 1) ** dim)
    lib().mmo_gll_coefficients(order, dim, xi, out)
    return out


def gll_inverse_transform(order, pnt, ctrl):
    pnt = np.ascontiguousarray(pnt, dtype=np.float64)
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    xi = np.zeros(pnt.shape[0])
    lib().mmo_gll_inverse_transform(order, pnt.shape[0], pnt, ctrl, xi)
    return xi


def set_gll_strict(on: bool) -> None:
    """Route the GLL inverse transform through the independent strict statement (plain sums, no fma, Gaussian
    solve, 1e-13): used by the tests to BOUND the production arithmetic, never by a parity comparison."""
    lib().mmo_set_gll_strict(1 if on else 0)


def locate_gll(order, nn, gll_points, points, tolerance=1.05, snap_to_nearest=False):
    """Control flow of reference interpolator.py:1181-1233 -> (elem int64[N], coeffs f64[N,P], nmissing)."""
    nn = np.ascontiguousarray(nn, dtype=np.int64)
    gll_points = np.ascontiguousarray(gll_points, dtype=np.float64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    nelem, P, dim = gll_points.shape
    n, k = nn.shape
    elem = np.zeros(n, dtype=np.int64)
    coeffs = np.zeros((n, P))
    miss = lib().mmo_locate_gll(order, dim, k, n, nn, gll_points, nelem, points, tolerance,
                                1 if snap_to_nearest else 0, elem, coeffs)
    return elem, coeffs, int(miss)


def locate_gll_v1(order, nn, gll_points, points):
    """Control flow of reference interpolator.py:1409-1473 (bounding-box pre-test variant) ->
    (elem int64[N], coeffs f64[N,P], number of points whose final transform was NaN)."""
    nn = np.ascontiguousarray(nn, dtype=np.int64)
    gll_points = np.ascontiguousarray(gll_points, dtype=np.float64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    nelem, P, dim = gll_points.shape
    n, k = nn.shape
    elem = np.zeros(n, dtype=np.int64)
    coeffs = np.zeros((n, P))
    hard = lib().mmo_locate_gll_v1(order, dim, k, n, nn, gll_points, nelem, points, elem, coeffs)
    return elem, coeffs, int(hard)


def gather_elem(fields, elem, coeffs, point_major=True):
    """fields f64[C, E, P] (or [E, P]) -> np.sum(coeffs * field[elem], axis=1) per component."""
    fields = np.ascontiguousarray(fields, dtype=np.float64)
    if fields.ndim == 2:
        fields = fields[None]
    ncomp, nelem, P = fields.shape
    elem = np.ascontiguousarray(elem, dtype=np.int64)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    n = elem.shape[0]
    out = np.zeros((n, ncomp) if point_major else (ncomp, n))
    rc = lib().mmo_gather_elem(fields, nelem, ncomp, elem, coeffs, n, P, out, 1 if point_major else 0)
    if rc != 0:
        raise ValueError("P > 128 unsupported")
    return out


# ----------------------------------------------------------------------------- third party
def knn_ckdtree(src, pts, k, workers=1):
    """The reference's kNN: cKDTree(src, balanced_tree=False).query(pts, k) (cli.py:66-73)."""
    from scipy.spatial import cKDTree

    tree = cKDTree(src, balanced_tree=False)
    dist, idx = tree.query(pts, k=k, workers=workers)
    if k == 1:
        dist, idx = dist[:, None], idx[:, None]
    return np.ascontiguousarray(idx, dtype=np.int64), dist


def gather_numpy(fields, ids, w):
    """Literal NumPy statement of scripts/cli.py:100, per component -> f64[N,C]."""
    fields = np.atleast_2d(fields)
    return np.stack([np.sum(f[ids] * w, axis=1) for f in fields], axis=1)


# ----------------------------------------------------------------------------- reference
def ref_centroid(conn, points):
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    out = np.zeros((conn.shape[0], points.shape[1]))
    reference_lib().centroid(points.shape[1], conn.shape[0], conn.shape[1], conn, points, out)
    return out


def ref_locate_hex8(nn, conn, nodes, points):
    """Compiled reference triLinearInterpolator, driven as scripts/cli.py:76-95.

    The reference prints one "not any" line per failed point to C stdout; that
    output is redirected to /dev/null for the duration of the call.
    """
    nn = np.ascontiguousarray(nn, dtype=np.int64)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    n, k = nn.shape
    enc = np.zeros((n, 8), dtype=np.int64)
    w = np.zeros((n, 8))
    L = reference_lib()
    libc = C.CDLL(None)
    libc.fflush(None)
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        nf = L.triLinearInterpolator(k, n, nn, conn, enc, nodes, w, points)
        libc.fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    return enc, w, int(nf)
