"""Pin the CPU oracle (oracle/mm_oracle.c) before anything is checked against it.

(1) against the committed golden fixtures, whose expected outputs were produced by the
    reference's own compiled C + scipy.cKDTree + NumPy (tests/golden/make_golden.py);
(2) live against oracle/_ref/multi_mesh_ref.so when that build is present, on random sweeps
    that reach every branch of the locator (accept, last-candidate fallback, failure).
All comparisons are bit-exact.
"""
import numpy as np
import pytest

from multimesh_amd import synth
from oracle import oracle as O

HEX_CASES = ["hex8_small", "hex8_hard_k1", "hex8_hard_k3", "hex8_hard_k20", "hex8_structured"]


@pytest.mark.parametrize("name", HEX_CASES)
def test_oracle_matches_reference_fixture(golden, name):
    d = golden(name)
    k = int(d["k"])
    cen = O.centroid(d["conn_a"], d["points_a"])
    assert np.array_equal(cen, d["centroid"])
    assert np.array_equal(synth.reorder_hex8(d["conn_a"]), d["conn_reordered"])
    enc, w, nf, st = O.locate_hex8(d["nn"], d["conn_reordered"], d["points_a"], d["points_b"],
                                   want_status=True)
    assert nf == int(d["nfailed"])
    assert np.array_equal(enc, d["enc"])
    assert np.array_equal(w, d["w"])
    assert np.array_equal(O.gather(d["fields"], enc, w), d["values"])
    # failed rows stay zero (caller zero-initialises, reference cli.py:77-78)
    assert int((st < 0).sum()) == nf
    assert not w[st < 0].any() and not enc[st < 0].any()
    if name.startswith("hex8_hard"):
        assert (st >= k).sum() > 50, "fixture must exercise the last-candidate fallback"
        assert nf > 50, "fixture must exercise failures"


def test_fixture_properties(golden):
    d = golden("hex8_small")
    assert int(d["nfailed"]) == 0
    assert np.abs(d["w"].sum(1) - 1).max() < 1e-14          # partition of unity
    err = np.abs(d["values"][:, 0] - synth.field_linear(d["points_b"])).max()
    assert err < 1e-7                                        # trilinear field reproduced


@pytest.mark.parametrize("P", [4, 8, 25, 27, 125])
def test_gather_order_is_numpys(golden, P):
    d = golden("gather")
    v = O.gather(d[f"field_P{P}"], d[f"ids_P{P}"], d[f"w_P{P}"])
    assert np.array_equal(v, d[f"values_P{P}"])
    # and against NumPy live (the statement at reference cli.py:100)
    rng = np.random.default_rng(P)
    f = rng.normal(size=(1, 300))
    ids = rng.integers(0, 300, size=(200, P))
    w = rng.normal(size=(200, P))
    assert np.array_equal(O.gather(f, ids, w), O.gather_numpy(f, ids, w))
    assert np.array_equal(O.gather(f, ids, w, point_major=False)[0], O.gather_numpy(f, ids, w)[:, 0])
    # rows of a failed point (ids 0, weights 0.0 -- the caller's zero-initialised arrays, reference
    # cli.py:77-78) over negative field values: every product is -0.0, NumPy's sum is +0.0 because
    # the reduction starts from the identity
    fneg = -1.0 - np.abs(f)
    ids0, w0 = np.zeros((5, P), np.int64), np.zeros((5, P))
    want = O.gather_numpy(fneg, ids0, w0)
    got = O.gather(fneg, ids0, w0)
    assert not np.signbit(want).any() and np.array_equal(np.signbit(got), np.signbit(want))


def test_knn_brute_matches_ckdtree_fixture(golden):
    d = golden("knn")
    for k in (1, 5, 20):
        idx, d2 = O.knn_brute(d["src3"], d["q3"], k, want_d2=True)
        assert np.array_equal(idx, d[f"idx3_k{k}"])
        assert np.array_equal(np.sqrt(d2), d[f"dist3_k{k}"])
    assert np.array_equal(O.knn_brute(d["srcg"], d["qg"], 20), d["idxg_k20"])
    assert np.array_equal(O.knn_brute(d["src2"], d["q2"], 20), d["idx2_k20"])
    idx, d2 = O.knn_brute(d["srcs"], d["qs"], 20, want_d2=True)
    assert np.array_equal(idx, d["idxs_k20"])               # padded with nsrc
    assert np.array_equal(np.sqrt(d2), d["dists_k20"])      # padded with inf


def test_knn_brute_matches_ckdtree_live():
    rng = np.random.default_rng(0)
    src = rng.uniform(size=(1500, 3))
    q = rng.uniform(-0.1, 1.1, size=(300, 3))
    idx, _ = O.knn_ckdtree(src, q, 20)
    assert np.array_equal(O.knn_brute(src, q, 20), idx)


def test_empty_inputs():
    pa, ca = synth.hex_mesh(4)
    conn = synth.reorder_hex8(ca)
    enc, w, nf = O.locate_hex8(np.zeros((0, 20), np.int64), conn, pa, np.zeros((0, 3)))
    assert enc.shape == (0, 8) and nf == 0
    assert O.gather(np.ones((1, 10)), np.zeros((0, 8), np.int64), np.zeros((0, 8))).shape == (0, 1)
    assert O.centroid(np.zeros((0, 8), np.int64), pa).shape == (0, 3)


needs_ref = pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("seed,jitter,k", [(0, 0.2, 20), (1, 0.42, 2), (2, 0.45, 6), (3, 0.0, 8)])
def test_oracle_vs_compiled_reference_live(seed, jitter, k):
    pa, ca = synth.hex_mesh(9, seed=seed, jitter=jitter)
    rng = np.random.default_rng(100 + seed)
    pb = rng.uniform(-0.1, 1.1, size=(3000, 3))
    cen = O.centroid(ca, pa)
    assert np.array_equal(cen, O.ref_centroid(ca, pa))
    nn, _ = O.knn_ckdtree(cen, pb, k)
    conn = synth.reorder_hex8(ca)
    enc, w, nf = O.locate_hex8(nn, conn, pa, pb)
    enc_r, w_r, nf_r = O.ref_locate_hex8(nn, conn, pa, pb)
    assert nf == nf_r
    assert np.array_equal(enc, enc_r)
    assert np.array_equal(w, w_r)


@needs_ref
def test_newton_and_weights_unit_level_vs_reference():
    # unit-level goldens: the reference's helpers are individually callable
    R = O.reference_lib()
    L = O.lib()
    rng = np.random.default_rng(42)
    pa, ca = synth.hex_mesh(4, seed=5, jitter=0.4)
    conn = synth.reorder_hex8(ca)
    for _ in range(400):
        e = rng.integers(0, len(conn))
        vtx = np.ascontiguousarray(pa[conn[e]])
        p = vtx.mean(0) + rng.normal(scale=0.3, size=3)
        xi_r, xi_o = np.zeros(3), np.zeros(3)
        ok_r = R.inverseCoordinateTransform(p, vtx, xi_r)
        ok_o = L.mmo_hex8_newton(p, vtx, xi_o, None)
        assert ok_r == ok_o
        assert np.array_equal(xi_r, xi_o, equal_nan=True)
        if ok_r:
            w_r, w_o = np.zeros(8), np.zeros(8)
            R.interpolateAtPoint(xi_r, w_r)
            L.mmo_hex8_weights(xi_o, w_o)
            assert np.array_equal(w_r, w_o)
