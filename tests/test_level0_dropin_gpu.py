"""Level 0 of INTEGRATION.md -- "replace the .so, change nothing": the two legacy symbols bound with the
reference's OWN ctypes declarations (reference multi_mesh/helpers.py:43-81: ``c_int`` scalars although the C
signature says ``long long``, ``restype = c_void_p`` for the void function ``centroid``) and driven the way
reference scripts/cli.py:62-100 drives them, against the golden fixtures generated from the reference's
compiled C.  Bit for bit: node ids, weights, failed counts, interpolated values.

A second handle on the library is opened for this (``ctypes.CDLL`` on the same file): the declarations of
``multimesh_amd.helpers.load_lib`` (``c_int64`` scalars) must not leak into the test.
"""
import ctypes as C
import glob
import os
import time

import numpy as np
import pytest

from multimesh_amd import helpers

pytestmark = pytest.mark.gpu


def reference_style_lib():
    """What reference helpers.py:27-81 does, statement for statement, on this package's lib/ directory."""
    possible_files = glob.glob(os.path.join(helpers.LIB_DIR, "multi_mesh*.so"))
    assert possible_files, "multi_mesh_hip.so has not been built"
    lib = C.CDLL(sorted(possible_files)[0])
    i64_2d = np.ctypeslib.ndpointer(dtype=np.int64, ndim=2, flags=["C_CONTIGUOUS"])
    f64_2d = np.ctypeslib.ndpointer(dtype=np.float64, ndim=2, flags=["C_CONTIGUOUS"])
    lib.centroid.restype = C.c_void_p
    lib.centroid.argtypes = [C.c_int, C.c_int, C.c_int, i64_2d, f64_2d, f64_2d]
    lib.triLinearInterpolator.restype = C.c_int64
    lib.triLinearInterpolator.argtypes = [C.c_int, C.c_int, i64_2d, i64_2d, i64_2d, f64_2d, f64_2d, f64_2d]
    return lib


@pytest.mark.parametrize("name", ["hex8_small", "hex8_hard_k20", "hex8_hard_k3", "hex8_hard_k1"])
def test_cli_flow_with_the_references_own_declarations(golden, name):
    from scipy.spatial import cKDTree

    d = golden(name)
    lib = reference_style_lib()
    points_a, conn_a, points_b = d["points_a"], d["conn_a"], d["points_b"]
    # exodus.get_element_centroid -> lib.centroid (reference io/exodus.py; ndim, nelem, nodes per element)
    a_centroids = np.zeros((conn_a.shape[0], 3))
    lib.centroid(3, conn_a.shape[0], 8, np.ascontiguousarray(conn_a), np.ascontiguousarray(points_a), a_centroids)
    assert np.array_equal(a_centroids, d["centroid"])
    # scripts/cli.py:65-73
    centroid_tree = cKDTree(a_centroids, balanced_tree=False)
    nelem_to_search = int(d["k"])
    _, nearest_element_indices = centroid_tree.query(points_b, k=nelem_to_search)
    nearest_element_indices = np.ascontiguousarray(nearest_element_indices.reshape(len(points_b), nelem_to_search))
    assert np.array_equal(nearest_element_indices, d["nn"])
    # scripts/cli.py:76-95
    npoints = points_b.shape[0]
    enclosing_elem_node_indices = np.zeros((npoints, 8), dtype=np.int64)
    weights = np.zeros((npoints, 8))
    permutation = [0, 3, 2, 1, 4, 5, 6, 7]
    i = np.argsort(permutation)
    connectivity_reordered = conn_a[:, i]
    nfailed = lib.triLinearInterpolator(
        nelem_to_search,
        npoints,
        nearest_element_indices,
        np.ascontiguousarray(connectivity_reordered),
        enclosing_elem_node_indices,
        np.ascontiguousarray(points_a),
        weights,
        np.ascontiguousarray(points_b),
    )
    assert nfailed == int(d["nfailed"])
    assert np.array_equal(enclosing_elem_node_indices, d["enc"])
    assert np.array_equal(weights, d["w"])
    # scripts/cli.py:98-100
    for c, param_a in enumerate(d["fields"]):
        values = np.sum(param_a[enclosing_elem_node_indices] * weights, axis=1)
        assert np.array_equal(values, d["values"][:, c])


def test_125_consecutive_calls_like_exodus_2_gll(golden, capsys):
    """reference scripts/cli.py:183-195 calls triLinearInterpolator once per GLL point of the elements
    (125 times for order 4) on the same source mesh: every call must give the same rows, and the loop must
    not pay six device allocations per call (the device copies come from the context's grow-only cache)."""
    from multimesh_amd import synth
    from scipy.spatial import cKDTree

    lib = reference_style_lib()
    pa, ca = synth.hex_mesh(41, seed=1)
    conn = np.ascontiguousarray(synth.reorder_hex8(ca))
    cen = np.zeros((ca.shape[0], 3))
    lib.centroid(3, ca.shape[0], 8, ca, pa, cen)
    tree = cKDTree(cen, balanced_tree=False)
    rng = np.random.default_rng(5)
    pts = np.ascontiguousarray(rng.uniform(0.02, 0.98, size=(20_000, 3)))
    _, nn = tree.query(pts, k=20)
    nn = np.ascontiguousarray(nn)
    first = None
    t0 = time.perf_counter()
    for call in range(125):
        enc = np.zeros((len(pts), 8), dtype=np.int64)
        w = np.zeros((len(pts), 8))
        assert lib.triLinearInterpolator(20, len(pts), nn, conn, enc, pa, w, pts) == 0
        if first is None:
            first = (enc, w)
            t0 = time.perf_counter()        # the first call allocates
        else:
            assert np.array_equal(enc, first[0]) and np.array_equal(w, first[1])
    per_call_ms = (time.perf_counter() - t0) / 124 * 1e3
    with capsys.disabled():
        print(f"\n[level 0] 124 warm triLinearInterpolator calls (68,921 nodes, 20,000 points): {per_call_ms:.2f} ms per call")
    assert np.abs(first[1].sum(axis=1) - 1.0).max() < 1e-12
