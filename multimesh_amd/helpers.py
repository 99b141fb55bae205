"""Library loading helper -- the drop-in for reference ``multi_mesh/helpers.py:22-84``.

Same contract as the reference's ``load_lib()``: glob ``<package>/lib/multi_mesh*.so``, open it
with ``ctypes.CDLL``, declare the argument types of the two legacy symbols (``centroid``,
``triLinearInterpolator``), cache the handle in a module-level list, and raise ``ValueError``
when no library is found.  Differences, on purpose:

* scalar arguments are declared ``c_int64`` (the C signature is ``long long``; the reference's
  ``c_int`` only works by accident of the x86-64 calling convention);
* the ``mm_*`` device-pointer entry points of ``include/multimesh_hip.h`` are declared too;
* there is NO CPU fallback: the library is HIP code for gfx950 and every compute call fails
  (``MultiMeshHipError``) when no GPU is usable.
"""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np

#: where the built library lives: <this package>/lib (same place the reference looks, helpers.py:22-27)
LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
#: the opened library, kept for the life of the process (the reference keeps its handle the same way)
cache = []

MM_OK = 0
MM_KNN_MAX_K = 64
MM_FP_EXACT = 0
MM_FP_TOL = 1
STAGES = ("centroid", "knn_build", "knn_query", "locate", "gather", "knn_cell", "locate_pass0")

#: every symbol include/multimesh_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "centroid", "triLinearInterpolator",
    "mm_device_count", "mm_last_error", "mm_last_status",
    "mm_context_create", "mm_context_destroy", "mm_synchronize",
    "mm_device_alloc", "mm_device_free", "mm_copy_h2d", "mm_copy_d2h", "mm_memset",
    "mm_centroid", "mm_knn_build", "mm_knn_query", "mm_knn_destroy",
    "mm_locate_hex8", "mm_gather", "mm_interpolate_hex8", "mm_interpolate_hex8_host", "mm_locate_gll", "mm_gather_elem",
    "mm_scatter_elements", "mm_fluid_solid_fix", "mm_set_profiling", "mm_last_timings", "mm_set_lazy_lists", "mm_unique_points", "mm_locate_gll_bbox", "mm_interpolate_gll",
    "mm_set_fp_mode", "mm_get_fp_mode", "mm_last_locate_stats",
    "mm_source_create", "mm_source_destroy", "mm_interpolate_hex8_on", "mm_points_to_elements", "mm_unique_points_any_order",
)


class MultiMeshHipError(RuntimeError):
    """A negative MM_ERR_* code came back from multi_mesh_hip.so."""


def _i64_2d():
    return np.ctypeslib.ndpointer(dtype=np.int64, ndim=2, flags=["C_CONTIGUOUS"])


def _f64_2d():
    return np.ctypeslib.ndpointer(dtype=np.float64, ndim=2, flags=["C_CONTIGUOUS"])


def load_lib():
    if cache:
        return cache[0]
    # any file called multi_mesh*.so in lib/ qualifies, the first in sorted order wins (the contract of
    # the reference loader: same glob, same exception type when nothing is there)
    candidates = sorted(glob.glob(os.path.join(LIB_DIR, "multi_mesh*.so")))
    if not candidates:
        raise ValueError(
            f"no multi_mesh*.so under {LIB_DIR}: build multi_mesh_hip.so with `make -C multimesh_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`"
        )
    lib = C.CDLL(candidates[0])

    # ---- legacy symbols (reference helpers.py:43-81) ----
    lib.centroid.restype = None
    lib.centroid.argtypes = [C.c_int64, C.c_int64, C.c_int64, _i64_2d(), _f64_2d(), _f64_2d()]
    lib.triLinearInterpolator.restype = C.c_int64
    lib.triLinearInterpolator.argtypes = [
        C.c_int64, C.c_int64, _i64_2d(), _i64_2d(), _i64_2d(), _f64_2d(), _f64_2d(), _f64_2d(),
    ]

    # ---- device-pointer API (include/multimesh_hip.h) ----
    vp = C.c_void_p
    lib.mm_device_count.restype = C.c_int
    lib.mm_device_count.argtypes = []
    lib.mm_last_error.restype = C.c_char_p
    lib.mm_last_error.argtypes = []
    lib.mm_last_status.restype = C.c_int
    lib.mm_last_status.argtypes = []
    lib.mm_context_create.restype = C.c_int
    lib.mm_context_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.mm_context_destroy.restype = None
    lib.mm_context_destroy.argtypes = [vp]
    lib.mm_synchronize.restype = C.c_int
    lib.mm_synchronize.argtypes = [vp]
    lib.mm_device_alloc.restype = C.c_int
    lib.mm_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.mm_device_free.restype = C.c_int
    lib.mm_device_free.argtypes = [vp, vp]
    lib.mm_copy_h2d.restype = C.c_int
    lib.mm_copy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mm_copy_d2h.restype = C.c_int
    lib.mm_copy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mm_memset.restype = C.c_int
    lib.mm_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    lib.mm_centroid.restype = C.c_int
    lib.mm_centroid.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp]
    lib.mm_knn_build.restype = C.c_int
    lib.mm_knn_build.argtypes = [vp, vp, C.c_int64, C.c_int64, C.POINTER(vp)]
    lib.mm_knn_query.restype = C.c_int
    lib.mm_knn_query.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, vp, vp]
    lib.mm_knn_destroy.restype = None
    lib.mm_knn_destroy.argtypes = [vp, vp]
    lib.mm_locate_hex8.restype = C.c_int64
    lib.mm_locate_hex8.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int, vp, vp, vp, vp]
    lib.mm_gather.restype = C.c_int
    lib.mm_gather.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, C.c_int]
    lib.mm_scatter_elements.restype = C.c_int
    lib.mm_scatter_elements.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int64, C.c_int64, vp]
    lib.mm_fluid_solid_fix.restype = C.c_int64
    lib.mm_fluid_solid_fix.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    lib.mm_interpolate_hex8.restype = C.c_int64
    lib.mm_interpolate_hex8.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                        C.c_int64, vp, vp, vp]
    lib.mm_interpolate_hex8_host.restype = C.c_int64
    lib.mm_interpolate_hex8_host.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                        C.c_int64, vp, vp, vp]
    lib.mm_locate_gll.restype = C.c_int64
    lib.mm_locate_gll.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp, C.c_int64, vp, C.c_double,
                                  C.c_int, vp, vp]
    lib.mm_gather_elem.restype = C.c_int
    lib.mm_gather_elem.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, C.c_int64, vp, C.c_int]
    lib.mm_interpolate_gll.restype = C.c_int64
    lib.mm_interpolate_gll.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int64,
                                       C.c_double, C.c_int, vp, vp, vp]
    lib.mm_locate_gll_bbox.restype = C.c_int64
    lib.mm_locate_gll_bbox.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp, C.c_int64, vp, vp, vp]
    lib.mm_unique_points.restype = C.c_int64
    lib.mm_unique_points.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp]
    lib.mm_set_lazy_lists.restype = C.c_int
    lib.mm_set_lazy_lists.argtypes = [vp, C.c_int]
    lib.mm_unique_points_any_order.restype = C.c_int64
    lib.mm_unique_points_any_order.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp]
    lib.mm_points_to_elements.restype = C.c_int
    lib.mm_points_to_elements.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.mm_source_create.restype = C.c_int
    lib.mm_source_create.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(vp)]
    lib.mm_source_destroy.restype = None
    lib.mm_source_destroy.argtypes = [vp, vp]
    lib.mm_interpolate_hex8_on.restype = C.c_int64
    lib.mm_interpolate_hex8_on.argtypes = [vp, vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, vp, vp, vp]
    lib.mm_set_fp_mode.restype = C.c_int
    lib.mm_set_fp_mode.argtypes = [vp, C.c_int]
    lib.mm_get_fp_mode.restype = C.c_int
    lib.mm_get_fp_mode.argtypes = [vp]
    lib.mm_last_locate_stats.restype = C.c_int
    lib.mm_last_locate_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.mm_set_profiling.restype = C.c_int
    lib.mm_set_profiling.argtypes = [vp, C.c_int]
    lib.mm_last_timings.restype = C.c_int
    lib.mm_last_timings.argtypes = [vp, C.POINTER(C.c_double), C.c_int]

    lib._filename = candidates[0]
    cache.append(lib)
    return lib


def check(rc, what="multi_mesh_hip call"):
    """Raise on a negative return code; pass non-negative values (counts) through."""
    if rc < 0:
        msg = load_lib().mm_last_error().decode(errors="replace")
        raise MultiMeshHipError(f"{what} failed with code {rc}: {msg}")
    return rc
