"""GPU parity tests proper: every call goes through the C ABI of multi_mesh_hip.so and is compared
bit-for-bit with (a) the committed golden fixtures generated from the reference itself and (b) the
CPU oracle on the same seeded inputs; at BASELINE.json's full sizes, with size-independent
properties plus an oracle check on a random sample of targets.

Tolerances: integer outputs (neighbour indices, node ids, failed counts) exact; floating-point
outputs (centroids, weights, distances, gathered values) also exact (0 ulp) -- the kernels keep the
reference's operation order and are built without fused multiply-add.  The only tolerance in this
file is for the tie-laden structured mesh, where the reference's own kNN order is unspecified.
"""
import numpy as np
import pytest

from multimesh_amd import helpers, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

HEX_CASES = ["hex8_small", "hex8_hard_k1", "hex8_hard_k3", "hex8_hard_k20", "hex8_structured"]


@pytest.fixture(scope="module")
def ctx():
    from multimesh_amd.device import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def lib():
    return helpers.load_lib()


# ------------------------------------------------------------------------------- A1 centroid
@pytest.mark.parametrize("name", HEX_CASES)
def test_centroid_golden(ctx, lib, golden, name):
    d = golden(name)
    conn, pts = np.ascontiguousarray(d["conn_a"]), np.ascontiguousarray(d["points_a"])
    out = np.zeros((conn.shape[0], 3))
    lib.centroid(3, conn.shape[0], 8, conn, pts, out)          # legacy symbol, host arrays
    assert lib.mm_last_status() == 0
    assert np.array_equal(out, d["centroid"])
    assert np.array_equal(ctx.centroid(conn, pts).numpy(), d["centroid"])  # device API


def test_centroid_2d_and_generic(ctx):
    p2, c2 = synth.quad_mesh(40, seed=3)
    assert np.array_equal(ctx.centroid(c2, p2).numpy(), O.centroid(c2, p2))
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(500, 3))
    conn = rng.integers(0, 500, size=(300, 27))               # order-2 hex: 27 nodes per element
    assert np.array_equal(ctx.centroid(conn, pts).numpy(), O.centroid(conn, pts))
    assert ctx.centroid(np.zeros((0, 8), np.int64), pts).numpy().shape == (0, 3)


# ------------------------------------------------------------------------------- A2 kNN
def test_knn_golden(ctx, golden):
    d = golden("knn")
    tree = ctx.knn_build(d["src3"])
    for k in (1, 5, 20):
        idx, dist = tree.query(d["q3"], k, want_dist=True)
        assert np.array_equal(idx.numpy(), d[f"idx3_k{k}"])
        assert np.array_equal(dist.numpy(), d[f"dist3_k{k}"])
    idx = ctx.knn_build(d["srcg"]).query(d["qg"], 20)          # graded density: ring expansion
    assert np.array_equal(idx.numpy(), d["idxg_k20"])
    idx = ctx.knn_build(d["src2"]).query(d["q2"], 20)          # 2-D
    assert np.array_equal(idx.numpy(), d["idx2_k20"])
    idx, dist = ctx.knn_build(d["srcs"]).query(d["qs"], 20, want_dist=True)   # nsrc < k: padded
    assert np.array_equal(idx.numpy(), d["idxs_k20"])
    assert np.array_equal(dist.numpy(), d["dists_k20"])


@pytest.mark.parametrize("k", [3, 20, 25, 30, 64])
def test_knn_vs_ckdtree_live(ctx, k):
    rng = np.random.default_rng(k)
    src = rng.uniform(size=(200_000, 3))
    q = rng.uniform(-0.05, 1.05, size=(50_000, 3))
    idx = ctx.knn_build(src).query(q, k).numpy()
    ref, _ = O.knn_ckdtree(src, q, k, workers=-1)
    assert np.array_equal(idx, ref)


def test_knn_structured_ties_are_index_ordered(ctx):
    # exact ties: cKDTree's order is unspecified; ours is (distance, index) like the brute oracle
    pa, ca = synth.hex_mesh(9, jitter=0.0)
    cen = O.centroid(ca, pa)
    idx = ctx.knn_build(cen).query(pa, 20).numpy()
    assert np.array_equal(idx, O.knn_brute(cen, pa, 20))


def test_knn_anisotropic_and_degenerate_clouds(ctx):
    rng = np.random.default_rng(4)
    flat = np.concatenate([rng.uniform(size=(5000, 2)), np.zeros((5000, 1))], axis=1)   # zero extent in z
    q = rng.uniform(-0.2, 1.2, size=(700, 3))
    assert np.array_equal(ctx.knn_build(flat).query(q, 20).numpy(), O.knn_ckdtree(flat, q, 20)[0])
    line = np.stack([np.linspace(0, 1, 4000) ** 3, np.zeros(4000), np.zeros(4000)], axis=1)  # graded 1-D
    assert np.array_equal(ctx.knn_build(line).query(q, 8).numpy(), O.knn_brute(line, q, 8))
    one = np.array([[0.3, 0.2, 0.1]])
    idx = ctx.knn_build(one).query(q[:10], 4).numpy()
    assert np.array_equal(idx, np.tile([0, 1, 1, 1], (10, 1)))
    assert ctx.knn_build(flat).query(np.zeros((0, 3)), 20).numpy().shape == (0, 20)


# ------------------------------------------------------------------------------- A4 locate
@pytest.mark.parametrize("name", HEX_CASES)
def test_locate_golden_legacy_symbol(lib, golden, name):
    d = golden(name)
    n, k = d["nn"].shape
    enc = np.zeros((n, 8), np.int64)
    w = np.zeros((n, 8))
    nf = lib.triLinearInterpolator(k, n, np.ascontiguousarray(d["nn"]), np.ascontiguousarray(d["conn_reordered"]),
                                   enc, np.ascontiguousarray(d["points_a"]), w, np.ascontiguousarray(d["points_b"]))
    assert nf == int(d["nfailed"])
    assert np.array_equal(enc, d["enc"])
    assert np.array_equal(w, d["w"])


@pytest.mark.parametrize("name", HEX_CASES)
def test_locate_golden_device_api(ctx, golden, name):
    d = golden(name)
    enc, w, nf = ctx.locate_hex8(d["nn"], d["conn_reordered"], d["points_a"], d["points_b"])
    assert nf == int(d["nfailed"])
    assert np.array_equal(enc.numpy(), d["enc"]) and np.array_equal(w.numpy(), d["w"])
    # the on-the-fly exodus reorder gives the same rows as the reference's host-side reorder
    enc2, w2, nf2 = ctx.locate_hex8(d["nn"], d["conn_a"], d["points_a"], d["points_b"], conn_is_exodus=True)
    assert nf2 == nf and np.array_equal(enc2.numpy(), d["enc"]) and np.array_equal(w2.numpy(), d["w"])


def test_locate_in_place_contract(ctx, golden):
    # rows of failed points keep the caller's contents (reference writes nothing for them)
    d = golden("hex8_hard_k3")
    n = d["nn"].shape[0]
    enc0 = np.full((n, 8), 7, np.int64)
    w0 = np.full((n, 8), 0.25)
    enc, w, nf = ctx.locate_hex8(d["nn"], d["conn_reordered"], d["points_a"], d["points_b"],
                                 enc=ctx.to_device(enc0), weights=ctx.to_device(w0))
    failed = ~d["w"].any(axis=1)
    assert nf == failed.sum() == int(d["nfailed"])
    assert np.array_equal(enc.numpy()[failed], enc0[failed]) and np.array_equal(w.numpy()[failed], w0[failed])
    assert np.array_equal(enc.numpy()[~failed], d["enc"][~failed])


def test_locate_vs_oracle_medium(ctx):
    pa, ca = synth.hex_mesh(33, seed=21, jitter=0.3)
    rng = np.random.default_rng(22)
    pb = rng.uniform(-0.03, 1.03, size=(60_000, 3))
    conn = synth.reorder_hex8(ca)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc, w, nf = ctx.locate_hex8(nn, conn, pa, pb)
    enc_o, w_o, nf_o = O.locate_hex8(nn, conn, pa, pb)
    assert nf == nf_o and nf > 0
    assert np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o)


def test_locate_padding_guard_and_empty(ctx):
    pa, ca = synth.hex_mesh(3)                                  # 8 elements < k = 20
    conn = synth.reorder_hex8(ca)
    pb = np.random.default_rng(0).uniform(size=(50, 3))
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20)           # padded with index 8
    enc, w, nf = ctx.locate_hex8(nn, conn, pa, pb)
    enc_o, w_o, nf_o = O.locate_hex8(nn[:, :8], conn, pa, pb)   # the 8 real candidates
    assert nf == nf_o == 0 and np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o)
    e, ww, nf = ctx.locate_hex8(np.zeros((0, 20), np.int64), conn, pa, np.zeros((0, 3)))
    assert nf == 0 and e.numpy().shape == (0, 8)


# ------------------------------------------------------------------------------- A9 gather
@pytest.mark.parametrize("P", [4, 8, 25, 27, 125])
def test_gather_golden(ctx, golden, P):
    d = golden("gather")
    f, ids, w = d[f"field_P{P}"], d[f"ids_P{P}"], d[f"w_P{P}"]
    assert np.array_equal(ctx.gather(f, ids, w).numpy(), d[f"values_P{P}"])
    assert np.array_equal(ctx.gather(f, ids, w, point_major=False).numpy(), d[f"values_P{P}"].T)
    assert np.array_equal(ctx.gather(f[0], ids, w).numpy()[:, 0], d[f"values_P{P}"][:, 0])


@pytest.mark.parametrize("P", [4, 8, 27])
def test_gather_rows_of_failed_points_read_plus_zero(ctx, P):
    # ids 0 / weights 0.0 (the caller's zero-initialised rows of a failed point, reference cli.py:77-78)
    # over negative field values: every product is -0.0; NumPy's reduction starts from the identity
    f = -1.0 - np.abs(np.random.default_rng(P).normal(size=(2, 50)))
    ids, w = np.zeros((9, P), np.int64), np.zeros((9, P))
    want = O.gather_numpy(f, ids, w)
    got = ctx.gather(f, ids, w).numpy()
    assert np.array_equal(got, want) and not np.signbit(got).any()


def test_gather_ragged_sizes_vs_oracle(ctx):
    rng = np.random.default_rng(8)
    for n, P, C in [(1, 8, 1), (63, 8, 3), (65, 8, 2), (1000, 27, 3), (17, 125, 2), (5, 9, 1), (0, 8, 2)]:
        f = rng.normal(size=(C, 777))
        ids = rng.integers(0, 777, size=(n, P))
        w = rng.normal(size=(n, P))
        assert np.array_equal(ctx.gather(f, ids, w).numpy(), O.gather(f, ids, w))


# ------------------------------------------------------------------------------- fused path
@pytest.mark.parametrize("name", ["hex8_small", "hex8_hard_k1", "hex8_hard_k3", "hex8_hard_k20"])
def test_fused_pipeline_golden(ctx, golden, name):
    # general-position meshes: our kNN equals cKDTree's, so everything downstream is bit-equal
    d = golden(name)
    vals, enc, w, nf = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"],
                                            nelem_to_search=int(d["k"]), want_operator=True)
    assert nf == int(d["nfailed"])
    assert np.array_equal(enc.numpy(), d["enc"]) and np.array_equal(w.numpy(), d["w"])
    assert np.array_equal(vals.numpy(), d["values"])


@pytest.mark.parametrize("name", ["hex8_small", "hex8_hard_k1", "hex8_hard_k3", "hex8_hard_k20"])
def test_fused_pipeline_values_only_golden(ctx, golden, name):
    # values-only call: no operator rows are materialised, the weighted sum is formed inside the
    # locate kernels.  Same bits as NumPy's gather of the reference's rows, failed points (zero
    # rows) and the last-candidate fallback included -- compared as raw bytes so that a -0.0 counts.
    d = golden(name)
    vals, nf = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"],
                                    nelem_to_search=int(d["k"]))
    assert nf == int(d["nfailed"])
    assert vals.numpy().tobytes() == np.ascontiguousarray(d["values"]).tobytes()


@pytest.mark.parametrize("name", ["hex8_hard_k20", "hex8_small"])
def test_fused_pipeline_eager_lists_golden(ctx, golden, name):
    # the same fixtures with the lazy evaluation of the candidate lists switched off (the default,
    # lazy, is what every other fused test runs): all nelem_to_search candidates up front
    d = golden(name)
    ctx.set_lazy_lists(False)
    try:
        vals, enc, w, nf = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"],
                                                nelem_to_search=int(d["k"]), want_operator=True)
    finally:
        ctx.set_lazy_lists(True)
    assert nf == int(d["nfailed"])
    assert np.array_equal(enc.numpy(), d["enc"]) and np.array_equal(w.numpy(), d["w"])
    assert vals.numpy().tobytes() == np.ascontiguousarray(d["values"]).tobytes()


def test_fused_pipeline_lazy_equals_eager_when_lists_run_out(ctx):
    # a strongly sheared source mesh and targets partly outside it: many targets are not accepted
    # within their 8 nearest centroids (some in none of 20), so the on-demand full lists, the
    # reference-order fallback and the failure path all run.  Lazy and eager must agree bit for bit,
    # and both with the oracle.
    pa, ca = synth.hex_mesh(24, seed=5, jitter=0.3)
    pa = pa.copy()
    pa[:, 0] += 0.9 * pa[:, 2] + 0.5 * pa[:, 1]            # shear: nearest centroid != containing element
    pa[:, 2] *= 0.15                                       # flat elements
    rng = np.random.default_rng(11)
    pb = rng.uniform(pa.min(axis=0) - 0.02, pa.max(axis=0) + 0.02, size=(40_000, 3))
    fields = synth.vector_field(pa)[:2]
    out = {}
    for lazy in (True, False):
        ctx.set_lazy_lists(lazy)
        try:
            vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
        finally:
            ctx.set_lazy_lists(True)
        out[lazy] = (vals.numpy(), enc.numpy(), w.numpy(), nf)
    for a, b in zip(out[True], out[False]):
        assert np.array_equal(a, b)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o, status = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb, want_status=True)
    assert (status >= 8).sum() > 100                        # candidates beyond the 8th were really needed
    assert out[True][3] == nf_o and nf_o > 0
    ok = status >= 0
    assert np.array_equal(out[True][1][ok], enc_o[ok]) and np.array_equal(out[True][2][ok], w_o[ok])


def test_fused_pipeline_many_components_uses_the_gather_kernel(ctx):
    # more than 3 components: the values-only call writes the operator internally and streams it
    # through the gather kernel instead of gathering inside the locate; same bits either way
    pa, ca = synth.hex_mesh(30, seed=1)
    pb, _ = synth.hex_mesh(33, seed=7)
    base = synth.vector_field(pa)
    fields = np.ascontiguousarray(np.concatenate([base, base[::-1] * 0.5]))      # C = 6
    vals, nf = ctx.interpolate_hex8(pa, ca, pb, fields)
    assert nf == 0 and vals.shape == (len(pb), 6)
    vals3, _ = ctx.interpolate_hex8(pa, ca, pb, fields[:3])                      # fused path
    assert np.array_equal(vals.numpy()[:, :3], vals3.numpy())
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
    assert np.array_equal(vals.numpy(), O.gather(fields, enc_o, w_o))


def test_fused_pipeline_structured_ties(ctx, golden):
    # exact kNN ties: candidate order (hence the chosen element on shared faces) is unspecified in
    # the reference; the interpolated values still agree to rounding.  Tolerance: 1e-13 absolute.
    d = golden("hex8_structured")
    vals, nf = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"])
    assert nf == 0
    assert np.abs(vals.numpy() - d["values"]).max() < 1e-13


# ------------------------------------------------------------------------------- full size
def _full_size_check(ctx, n_src, n_tgt, ncomp, sample):
    pa, ca = synth.hex_mesh(n_src, seed=1)
    pb, _ = synth.hex_mesh(n_tgt, seed=7)
    fields = synth.vector_field(pa)[:ncomp]
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
    vals, enc, w = vals.numpy(), enc.numpy(), w.numpy()
    assert nf == 0                                              # all targets inside the hull
    assert np.abs(w.sum(axis=1) - 1).max() < 1e-13              # partition of unity
    assert np.abs(vals[:, 0] - synth.field_linear(pb)).max() < 1e-7   # trilinear field reproduced
    assert enc.min() >= 0 and enc.max() < len(pa)
    # idempotence / determinism: a second run is bit-identical
    vals2, nf2 = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
    assert nf2 == 0 and np.array_equal(vals2.numpy(), vals)
    # oracle on a random sample of targets (cKDTree over ALL source centroids)
    pick = np.sort(np.random.default_rng(3).choice(len(pb), size=sample, replace=False))
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb[pick], 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb[pick])
    assert nf_o == 0
    assert np.array_equal(enc[pick], enc_o) and np.array_equal(w[pick], w_o)
    assert np.array_equal(vals[pick], O.gather(fields, enc_o, w_o))


def test_cfg2_1M_to_1M(ctx):
    _full_size_check(ctx, 101, 101, 1, sample=50_000)


def test_cfg3_10M_to_10M_vector_field(ctx):
    _full_size_check(ctx, 216, 216, 3, sample=20_000)


@pytest.mark.parametrize("shard", [0, 7])
def test_cfg4_real_shards_of_the_100M_target_mesh(ctx, shard):
    # cfg4 as the 8-GPU run sees it: rank r's block = shard_bounds(465^3, 8, r) rows of the 465^3 target
    # mesh -- a 58-plane slab with ~10 targets per source node that touches 1/8 of the 216^3 source
    # grid (~160 targets per kNN strip: the strips-shared-between-waves regime), NOT a cube of its own.
    # Shard 0 and shard 7 (the short last one) on the one GPU; properties on all 12.57M targets,
    # cKDTree + oracle on a 20k sample.
    from multimesh_amd.distributed import shard_bounds

    n_tgt = synth.CONFIGS["cfg4"]["n_tgt"]
    lo, hi = shard_bounds(n_tgt ** 3, 8, shard)
    assert (lo, hi) == ((0, 12_568_079) if shard == 0 else (87_976_553, 100_544_625))
    pa, ca = synth.hex_mesh(synth.CONFIGS["cfg4"]["n_src"], seed=1)
    pb = synth.hex_mesh_rows(n_tgt, lo, hi, seed=7)
    fields = synth.vector_field(pa)[:1]
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
    vals, enc, w = vals.numpy(), enc.numpy(), w.numpy()
    assert nf == 0
    assert np.abs(w.sum(axis=1) - 1).max() < 1e-13
    assert np.abs(vals[:, 0] - synth.field_linear(pb)).max() < 1e-7
    v2, nf2 = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)      # values only (fused gather)
    assert nf2 == 0 and np.array_equal(v2.numpy(), vals)
    pick = np.sort(np.random.default_rng(4 + shard).choice(len(pb), size=20_000, replace=False))
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb[pick], 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb[pick])
    assert nf_o == 0
    assert np.array_equal(enc[pick], enc_o) and np.array_equal(w[pick], w_o)
    assert np.array_equal(vals[pick], O.gather(fields, enc_o, w_o))


def test_cfg5_full_size_order4_gll(ctx):
    # cfg5 at SURVEY section 8's size: 43^3 order-4 source elements (9.94M element-nodal points), targets =
    # the unique GLL points of the 47^3-element mesh (13.0M element-nodal -> 7.2M unique) through
    # mm_unique_points + the fused mm_interpolate_gll (reference flow interpolator.py:666-826).
    # GLL numerics are parity-unpinned (salvus.fem absent): the checks are the analytic properties on
    # ALL targets and HIP == oracle on a 10k sample.
    src = synth.gll_mesh(44, 4, seed=1)
    tgt = synth.gll_mesh(48, 4, seed=7)
    assert src.shape == (43 ** 3, 125, 3) and tgt.shape == (47 ** 3, 125, 3)
    uniq, inv = ctx.unique_points(tgt.reshape(-1, 3))
    U = uniq.shape[0]
    # 189^3 = 6.75M geometric points; copies of a shared face point computed from different elements
    # may differ in the last bit and then stay distinct, exactly as for np.unique in the reference
    assert (47 * 4 + 1) ** 3 <= U < 7_400_000
    pts = uniq.numpy()
    assert np.array_equal(pts[inv.numpy()[:100_000]], tgt.reshape(-1, 3)[:100_000])   # the inverse index scatters back
    fields = np.stack([synth.field_linear(src), np.ones(src.shape[:2])])
    vals, elem, co, miss = ctx.interpolate_gll(4, src, uniq, fields, nelem_to_search=20, tolerance=1.05,
                                               want_operator=True)
    assert miss == 0
    vals, elem = vals.numpy(), elem.numpy()
    assert elem.min() >= 0 and elem.max() < src.shape[0]
    assert np.abs(vals[:, 0] - synth.field_linear(pts)).max() < 1e-11     # linear field reproduced
    assert np.abs(vals[:, 1] - 1.0).max() < 1e-12                         # sum of coefficients = 1 (constant field)
    v2, miss2 = ctx.interpolate_gll(4, src, uniq, fields, nelem_to_search=20, tolerance=1.05)   # fused, no operator
    assert miss2 == 0 and np.array_equal(v2.numpy(), vals)
    pick = np.sort(np.random.default_rng(5).choice(U, size=10_000, replace=False))
    co_pick = np.stack([co.rows(int(i), int(i) + 1).numpy()[0] for i in pick[:200]])   # rows of the 6.7 GB operator
    nn, _ = O.knn_ckdtree(src.mean(axis=1), pts[pick], 20, workers=-1)
    elem_o, co_o, miss_o = O.locate_gll(4, nn, src, pts[pick], tolerance=1.05, snap_to_nearest=False)
    assert miss_o == 0
    assert np.array_equal(elem[pick], elem_o) and np.array_equal(co_pick, co_o[:200])
    assert np.array_equal(vals[pick], O.gather_elem(fields, elem_o, co_o))


@pytest.mark.parametrize("order", [1, 4])
def test_cfg1_full_size_2d_through_the_api(order):
    # cfg1 at its own size: 2-D 100x100-node meshes (99^2 quads), GLL -> GLL through the API's array core
    # (unique targets, locate, gather, scatter back: reference interpolator.py:666-826) against the
    # oracle on EVERY target
    from multimesh_amd import api

    src = synth.gll_mesh(100, order, seed=1, dim=2)
    tgt = synth.gll_mesh(100, order, seed=7, dim=2)
    f = synth.field_smooth(src.reshape(-1, 2)).reshape(src.shape[:2])
    mesh = api.GllMesh(src, order, {"f": f, "lin": synth.field_linear(src)})
    out = api.interpolate_gll_to_gll(mesh, tgt, ["f", "lin"], nelem_to_search=20, tolerance=1.05)
    assert out.shape == (2,) + tgt.shape[:2]
    uniq, inv = np.unique(tgt.reshape(-1, 2), axis=0, return_inverse=True)          # reference utils.py:484-488
    nn, _ = O.knn_ckdtree(src.mean(axis=1), uniq, 20)
    elem_o, co_o, miss_o = O.locate_gll(order, nn, src, uniq, tolerance=1.05, snap_to_nearest=False)
    assert miss_o == 0
    want = O.gather_elem(np.stack([f, synth.field_linear(src)]), elem_o, co_o)[inv.reshape(-1)]
    assert np.array_equal(out.reshape(2, -1).T, want)
    assert np.abs(out[1] - synth.field_linear(tgt)).max() < 1e-11


def test_host_array_entry_equals_the_resident_pipeline(ctx):
    # mm_interpolate_hex8_host (uploads overlapped with the kernels, device copies cached in the context):
    # same bits as the resident-array entry and the oracle, across calls of growing and shrinking size,
    # with and without the operator, with more components than the fused gather takes
    for n_src, n_tgt, ncomp in ((13, 17, 1), (31, 29, 5), (9, 40, 3), (31, 29, 2)):
        pa, ca = synth.hex_mesh(n_src, seed=1)
        pb, _ = synth.hex_mesh(n_tgt, seed=7)
        pb = np.concatenate([pb, [[1.7, 0.5, 0.5], [-0.4, 2.0, 0.1]]])          # two targets outside: they fail
        fields = np.stack([synth.field_linear(pa), synth.field_smooth(pa), synth.field_xyz(pa),
                           synth.field_linear(pa) ** 2, -synth.field_smooth(pa)])[:ncomp]
        nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20)
        enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
        vals_o = O.gather(fields, enc_o, w_o)
        v_dev, nf_dev = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
        v1, nf1 = ctx.interpolate_hex8_host(pa, ca, pb, fields, nelem_to_search=20)
        out = np.full((len(pb), ncomp), 123.0)
        v2, enc, w, nf2 = ctx.interpolate_hex8_host(pa, ca, pb, fields, nelem_to_search=20, want_operator=True, out=out)
        assert v2 is out and nf1 == nf2 == nf_dev == nf_o == 2
        assert np.array_equal(v1, vals_o) and np.array_equal(v2, vals_o) and np.array_equal(v_dev.numpy(), vals_o)
        assert np.array_equal(enc, enc_o) and np.array_equal(w, w_o)
    with pytest.raises(ValueError):
        ctx.interpolate_hex8_host(pa, ca, pb, fields[:, :5])


def test_graded_meshes_through_the_pipeline(ctx):
    # meshes refined towards a corner (node coordinates u -> u^2.2 per axis: element sizes span three
    # orders of magnitude, so the centroid cloud needs several density levels), source and target of
    # different resolution; lazy and eager lists against cKDTree + the oracle
    def warp(p):
        return p ** 2.2

    pa, ca = synth.hex_mesh(61, seed=1, jitter=0.1)
    pb, _ = synth.hex_mesh(75, seed=7, jitter=0.1)
    pa, pb = warp(pa), warp(pb)
    fields = synth.vector_field(pa)[:2]
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
    vals_o = O.gather(fields, enc_o, w_o)
    for lazy in (True, False):
        ctx.set_lazy_lists(lazy)
        try:
            vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
            v2, nf2 = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
        finally:
            ctx.set_lazy_lists(True)
        assert nf == nf_o == nf2
        assert np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o)
        assert np.array_equal(vals.numpy(), vals_o) and np.array_equal(v2.numpy(), vals_o)


def test_graded_mesh_with_a_long_list_of_exhausted_targets(ctx):
    # a graded mesh on which well over 32768 targets find no acceptance among their 8 lazily evaluated candidates
    # (elongated elements: the containing element's centroid is not among the nearest): their full lists come from
    # the tiled kNN kernels restricted to the list, a second launch of the locate pass kernel walks candidates
    # 8 .. 19, and only what is left goes through the reference-order kernel.  Bit for bit against the oracle.
    pa, ca = synth.hex_mesh(90, seed=3, jitter=0.1)
    pb, _ = synth.hex_mesh(90, seed=9, jitter=0.1)
    pa, pb = pa ** 2.2, pb ** 2.2
    fields = synth.vector_field(pa)[:1]
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    conn = synth.reorder_hex8(ca)
    enc_o, w_o, nf_o, status = O.locate_hex8(nn, conn, pa, pb, want_status=True)
    assert (status >= 8).sum() + (status < 0).sum() > 40_000          # accepted beyond the 8th candidate, fallback or failed
    vals_o = O.gather(fields, enc_o, w_o)
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
    assert nf == nf_o
    assert np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o) and np.array_equal(vals.numpy(), vals_o)
    v2, nf2 = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
    assert nf2 == nf_o and np.array_equal(v2.numpy(), vals_o)


# ------------------------------------------------------------------------------- A10 GLL (parity unpinned)
@pytest.mark.parametrize("order,dim", [(o, d) for o in (1, 2, 4) for d in (2, 3)])
def test_gll_locate_and_gather_equal_the_oracle(ctx, order, dim):
    # our own numerics (the reference's live in the absent salvus.fem): kernel == oracle bit for bit
    gp = synth.gll_mesh(7 if dim == 3 else 12, order, seed=4, jitter=0.25, dim=dim)
    rng = np.random.default_rng(order + dim)
    pts = rng.uniform(-0.03, 1.03, size=(3000, dim))                # some outside -> -1 / snapped
    cen = gp.mean(axis=1)
    k = min(25, gp.shape[0])
    nn = ctx.knn_build(cen).query(pts, k).numpy()
    assert np.array_equal(nn, O.knn_ckdtree(cen, pts, k)[0])
    fields = np.stack([synth.field_linear(gp), synth.field_smooth(gp.reshape(-1, dim)).reshape(gp.shape[:2])])
    for tol, snap in ((1.05, False), (1.03, False), (1.05, True)):
        elem, co, miss = ctx.locate_gll(order, nn, gp, pts, tolerance=tol, snap_to_nearest=snap)
        elem_o, co_o, miss_o = O.locate_gll(order, nn, gp, pts, tolerance=tol, snap_to_nearest=snap)
        assert miss == miss_o and (snap or miss > 0)
        assert np.array_equal(elem.numpy(), elem_o)
        assert np.array_equal(co.numpy(), co_o)
        vals = ctx.gather_elem(fields, elem, co).numpy()
        assert np.array_equal(vals, O.gather_elem(fields, elem_o, co_o))
        assert np.array_equal(ctx.gather_elem(fields, elem, co, point_major=False).numpy(), vals.T)
        found = elem_o >= 0
        assert np.abs(vals[found, 0] - synth.field_linear(pts[found])).max() < 1e-11 or snap


@pytest.mark.parametrize("order,dim", [(o, d) for o in (1, 2, 4) for d in (2, 3)])
def test_gll_fused_pipeline_equals_staged_calls_and_oracle(ctx, order, dim):
    # mm_interpolate_gll: centroids (NumPy mean order), kNN, acceptance loop, weighted sum formed at
    # the point of acceptance, candidate lists evaluated lazily -- against cKDTree + the oracle's
    # locate + the oracle's NumPy-order gather, values-only and with the operator, lazy and eager.
    gp = synth.gll_mesh(7 if dim == 3 else 12, order, seed=9, jitter=0.25, dim=dim)
    # shear the mesh so that a good share of the targets is NOT in one of its 8 nearest elements
    gp = gp.copy()
    gp[..., 0] += 1.7 * gp[..., 1]
    rng = np.random.default_rng(10 * order + dim)
    lo, hi = gp.reshape(-1, dim).min(axis=0), gp.reshape(-1, dim).max(axis=0)
    pts = rng.uniform(lo - 0.02, hi + 0.02, size=(6000, dim))         # the sheared box: many outside
    cen = gp.mean(axis=1)
    k = min(20, gp.shape[0])
    fields = np.stack([synth.field_linear(gp), synth.field_smooth(gp.reshape(-1, dim)).reshape(gp.shape[:2]),
                       -1.0 - synth.field_linear(gp) ** 2])
    nn = O.knn_ckdtree(cen, pts, k)[0]
    for tol, snap in ((1.05, False), (1.05, True)):
        elem_o, co_o, miss_o = O.locate_gll(order, nn, gp, pts, tolerance=tol, snap_to_nearest=snap)
        vals_o = O.gather_elem(fields, elem_o, co_o)
        beyond8 = int(((elem_o[:, None] != nn[:, :8]).all(axis=1) & (elem_o >= 0)).sum())
        assert snap or (miss_o > 0 and (k <= 8 or beyond8 > 0))
        for lazy in (True, False):
            ctx.set_lazy_lists(lazy)
            try:
                vals, miss = ctx.interpolate_gll(order, gp, pts, fields, nelem_to_search=k, tolerance=tol,
                                                 snap_to_nearest=snap)
                v2, elem, co, miss2 = ctx.interpolate_gll(order, gp, pts, fields, nelem_to_search=k, tolerance=tol,
                                                          snap_to_nearest=snap, want_operator=True)
            finally:
                ctx.set_lazy_lists(True)
            assert miss == miss_o == miss2
            assert np.array_equal(vals.numpy(), vals_o) and np.array_equal(v2.numpy(), vals_o)
            # the sign of the zeros of points that were not found follows NumPy's field[-1] * 0.0
            assert np.array_equal(np.signbit(vals.numpy()), np.signbit(vals_o))
            assert np.array_equal(elem.numpy(), elem_o) and np.array_equal(co.numpy(), co_o)
        # one component, and none
        v1, _ = ctx.interpolate_gll(order, gp, pts, fields[1], nelem_to_search=k, tolerance=tol, snap_to_nearest=snap)
        assert np.array_equal(v1.numpy()[:, 0], vals_o[:, 1])


def test_gll_fused_pipeline_on_a_graded_mesh(ctx):
    # control nodes warped u -> u^2 per axis: element sizes span two orders of magnitude, the centroid
    # cloud gets density levels, and the lazily evaluated lists are completed level-aware
    gp = synth.gll_mesh(17, 2, seed=3, jitter=0.05) ** 2.0                 # 16^3 order-2 elements
    rng = np.random.default_rng(8)
    pts = rng.uniform(size=(60_000, 3)) ** 2.0
    fields = np.stack([synth.field_linear(gp), synth.field_smooth(gp.reshape(-1, 3)).reshape(gp.shape[:2])])
    nn = O.knn_ckdtree(gp.mean(axis=1), pts, 20, workers=-1)[0]
    elem_o, co_o, miss_o = O.locate_gll(2, nn, gp, pts, tolerance=1.05, snap_to_nearest=False)
    vals_o = O.gather_elem(fields, elem_o, co_o)
    for lazy in (True, False):
        ctx.set_lazy_lists(lazy)
        try:
            vals, elem, co, miss = ctx.interpolate_gll(2, gp, pts, fields, nelem_to_search=20, want_operator=True)
            v2, miss2 = ctx.interpolate_gll(2, gp, pts, fields, nelem_to_search=20)
        finally:
            ctx.set_lazy_lists(True)
        assert miss == miss_o == miss2
        assert np.array_equal(elem.numpy(), elem_o) and np.array_equal(co.numpy(), co_o)
        assert np.array_equal(vals.numpy(), vals_o) and np.array_equal(v2.numpy(), vals_o)


def test_gll_api_and_cfg5_shaped_run(ctx):
    # order-4 hexes as in cfg5 (reduced size): targets = the unique GLL points of a second mesh
    from multimesh_amd import api

    src = synth.gll_mesh(9, 4, seed=1, dim=3)                       # 8^3 elements x 125 nodes
    tgt = np.unique(synth.gll_mesh(8, 4, seed=7, dim=3).reshape(-1, 3), axis=0)   # reference utils.py:484-488
    mesh = api.GllMesh(src, 4, {"VP": synth.field_smooth(src.reshape(-1, 3)).reshape(src.shape[:2]),
                                "RHO": synth.field_linear(src)})
    vals = api.interpolate_gll_to_points(mesh, tgt, ["VP", "RHO"], nelem_to_search=20, tolerance=1.05)
    assert vals.shape == (len(tgt), 2)
    assert np.abs(vals[:, 1] - synth.field_linear(tgt)).max() < 1e-11
    assert np.abs(vals[:, 0] - synth.field_smooth(tgt)).max() < 2e-3
    elems, coeffs = api.get_element_weights(src, 4, src.mean(axis=1), tgt, nelem_to_search=20)
    nn, _ = O.knn_ckdtree(src.mean(axis=1), tgt, 20)
    elem_o, co_o, _ = O.locate_gll(4, nn, src, tgt)
    assert np.array_equal(elems, elem_o) and np.array_equal(coeffs, co_o)


# ------------------------------------------------------------------------------- kNN regimes
def test_knn_many_targets_per_cell_multi_round(ctx):
    # ~60 targets per grid cell: several rounds per wave at the narrowest group split
    rng = np.random.default_rng(11)
    src = rng.uniform(size=(20_000, 3))
    q = rng.uniform(size=(150_000, 3))
    assert np.array_equal(ctx.knn_build(src).query(q, 20).numpy(), O.knn_ckdtree(src, q, 20, workers=-1)[0])


@pytest.mark.parametrize("order_by", ["random", "lexicographic"])
def test_knn_many_targets_over_few_cells_histogram_count(ctx, order_by):
    # >= 64 targets per cell of a grid of <= 16,384 cells: the targets' counting sort counts through an LDS histogram
    # per workgroup (cell_count_hist_kernel); lexicographic order (what np.unique leaves) is the contended case
    rng = np.random.default_rng(12)
    src = rng.uniform(size=(4_000, 3))
    q = rng.uniform(-0.02, 1.02, size=(300_000, 3))
    if order_by == "lexicographic":
        q = q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))]
    for k in (8, 20):
        assert np.array_equal(ctx.knn_build(src).query(q, k).numpy(), O.knn_ckdtree(src, q, k, workers=-1)[0])


@pytest.mark.parametrize("k", [8, 20])
def test_knn_strips_shared_between_waves(ctx, k):
    # ~1,400 targets per strip of two cells (the unique GLL points of a fine mesh over the centroids of a
    # coarse one look like this): each strip's targets are shared out between several waves
    rng = np.random.default_rng(13)
    src = rng.uniform(size=(6_000, 3))
    q = rng.uniform(size=(520_003, 3))
    assert np.array_equal(ctx.knn_build(src).query(q, k).numpy(), O.knn_ckdtree(src, q, k, workers=-1)[0])


def test_knn_few_targets_wide_groups(ctx):
    # far fewer targets than cells: one target per wave, all 64 lanes on it
    rng = np.random.default_rng(12)
    src = rng.uniform(size=(300_000, 3))
    q = rng.uniform(-0.1, 1.1, size=(500, 3))
    for k in (1, 7, 20, 32):
        assert np.array_equal(ctx.knn_build(src).query(q, k).numpy(), O.knn_ckdtree(src, q, k, workers=-1)[0])


def test_knn_clustered_sources_overflow_the_tile(ctx):
    # a tight cluster inside a sparse cloud: cells holding hundreds of sources (tile overflow ->
    # generic kernel) next to nearly empty ones (ring expansion)
    rng = np.random.default_rng(13)
    src = np.concatenate([rng.normal(0.3, 0.004, size=(30_000, 3)), rng.uniform(size=(30_000, 3))])
    q = np.concatenate([rng.normal(0.3, 0.01, size=(3_000, 3)), rng.uniform(size=(3_000, 3))])
    assert np.array_equal(ctx.knn_build(src).query(q, 20).numpy(), O.knn_ckdtree(src, q, 20, workers=-1)[0])


def test_knn_duplicate_sources_tie_overflow(ctx):
    # every source repeated 40 times: far more exact ties than the collect list holds; the result
    # must still be the (distance, index)-ordered truth
    rng = np.random.default_rng(14)
    base = rng.uniform(size=(600, 3))
    src = np.repeat(base, 40, axis=0)
    q = rng.uniform(size=(400, 3))
    assert np.array_equal(ctx.knn_build(src).query(q, 20).numpy(), O.knn_brute(src, q, 20))


def test_knn_2d_mesh_like_cfg1(ctx):
    # cfg1: 2-D 100x100-node quad meshes (the reference runs this one on the CPU through cKDTree)
    pa, ca = synth.quad_mesh(100, seed=1)
    pb, _ = synth.quad_mesh(100, seed=7)
    cen = O.centroid(ca, pa)
    assert np.array_equal(ctx.centroid(ca, pa).numpy(), cen)
    assert np.array_equal(ctx.knn_build(cen).query(pb, 20).numpy(), O.knn_ckdtree(cen, pb, 20)[0])


def test_fused_pipeline_random_order_and_outside_targets(ctx):
    # targets in random order (no spatial coherence) with a shell outside the hull
    pa, ca = synth.hex_mesh(31, seed=1)
    rng = np.random.default_rng(15)
    pb = rng.uniform(-0.04, 1.04, size=(120_000, 3))
    fields = synth.vector_field(pa)
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, want_operator=True)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
    assert nf == nf_o and nf > 0
    assert np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o)
    assert np.array_equal(vals.numpy(), O.gather(fields, enc_o, w_o))


# ------------------------------------------------------------------------------- GLL, variant 1
@pytest.mark.parametrize("order,dim", [(1, 3), (2, 3), (4, 3), (2, 2), (4, 2)])
def test_gll_bbox_variant_equals_the_oracle(ctx, order, dim):
    # the bounding-box acceptance loop (reference interpolator.py:1409-1473): targets inside, on the
    # hull and outside the mesh, so that the accept, first-inside-box, nearest-centre and constant-xi
    # branches all run; element ids and coefficients bit-equal to the oracle's restatement
    src = synth.gll_mesh(7 if dim == 3 else 14, order, seed=2, dim=dim)
    rng = np.random.default_rng(order * 10 + dim)
    pts = rng.uniform(-0.08, 1.08, size=(6000, dim))
    nn, _ = O.knn_ckdtree(src.mean(axis=1), pts, 12, workers=-1)
    elem, coeffs, hard = ctx.locate_gll_bbox(order, nn, src, pts)
    e_o, c_o, h_o = O.locate_gll_v1(order, nn, src, pts)
    assert hard == h_o
    assert np.array_equal(elem.numpy(), e_o)
    assert np.array_equal(coeffs.numpy(), c_o)
    inside = ((pts > 0.02) & (pts < 0.98)).all(axis=1)
    # interior points: the accepted element reproduces the point through its own basis
    rec = np.einsum("np,npd->nd", c_o[inside], src[e_o[inside]])
    assert np.abs(rec - pts[inside]).max() < 1e-9
    outside = ((pts < -0.03) | (pts > 1.03)).any(axis=1)
    assert outside.sum() > 100 and np.abs(c_o[outside].sum(axis=1) - 1).max() < 1e-12


@pytest.mark.parametrize("k", [1, 2, 24, 25])
def test_knn_list_capacities_that_are_not_a_multiple_of_four(ctx, k):
    # K = 1, 2 and 25 give list capacities of 9, 10 and 37: the last batch of four of the rank loop
    # reaches past them.  On a strongly anisotropic lattice the isotropic density estimate is poor,
    # many targets collect more candidates than the capacity (they are handed over), and their
    # clamped reads once pushed ranks past the row -- into the neighbouring target's ids (found by
    # tools/fuzz_pipeline.py, seed 424242 case 615).  Dense targets: several rounds per strip.
    pa, ca = synth.hex_mesh(29, seed=766496109, jitter=0.104)
    pa = pa * np.array([0.238, 4.876, 4.723])               # plate-like elements, 20 : 1
    pa[:, 0] += 0.0276 * pa[:, 1]
    pa += np.array([-563.96, 979.72, -667.06])
    cen = O.centroid(ca, pa)
    rng = np.random.default_rng(5)
    q = rng.uniform(pa.min(axis=0), pa.max(axis=0), size=(60_000, 3))
    tree = ctx.knn_build(cen)
    ref, _ = O.knn_ckdtree(cen, q, k, workers=-1)
    ref = ref.reshape(len(q), k)
    for _ in range(3):
        assert np.array_equal(tree.query(q, k).numpy().reshape(len(q), k), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("power,k", [(1.5, 20), (2.0, 8), (3.0, 20), (2.0, 40), (3.0, 64)])
def test_knn_graded_cloud_density_levels(ctx, power, k, monkeypatch):
    # coordinates = uniform^p: the density varies by orders of magnitude, so the build adds grids laid
    # out for denser regions and targets whose strip overflows the tile are passed down a level
    # (mm_knn.hip, "density levels").  Same lists with the levels, with one grid only, and from the
    # k-d tree.
    rng = np.random.default_rng(17)
    src = rng.uniform(size=(150_000, 3)) ** power
    q = rng.uniform(size=(40_000, 3)) ** power
    ref, dref = O.knn_ckdtree(src, q, k, workers=-1)
    for levels in ("5", "1"):
        monkeypatch.setenv("MM_KNN_LEVELS", levels)
        tree = ctx.knn_build(src)
        idx, dist = tree.query(q, k, want_dist=True)
        assert np.array_equal(idx.numpy().reshape(len(q), k), ref.reshape(len(q), k))
        np.testing.assert_allclose(dist.numpy().reshape(len(q), k), dref.reshape(len(q), k), rtol=1e-12, atol=0)


def test_knn_two_densities_like_a_locally_refined_mesh(ctx):
    # a coarse background and a region refined 3x per axis (27x the density): neither fits a grid laid out
    # for the mean; with density levels both are served by the tiled kernel
    rng = np.random.default_rng(23)
    coarse = rng.uniform(size=(120_000, 3))
    fine = 0.35 + 0.2 * rng.uniform(size=(120_000, 3))
    src = np.concatenate([coarse, fine])
    q = np.concatenate([rng.uniform(size=(30_000, 3)), 0.35 + 0.2 * rng.uniform(size=(30_000, 3))])
    for k in (8, 20):
        assert np.array_equal(ctx.knn_build(src).query(q, k).numpy(), O.knn_ckdtree(src, q, k, workers=-1)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("power,k", [(1.0, 8), (1.5, 20), (2.0, 8), (3.0, 20), (3.0, 3), (2.0, 40)])
def test_knn_tree_on_graded_clouds(ctx, power, k, monkeypatch):
    # the density-adaptive index (mm_knn_tree.inc.h: Morton-sorted sources, binary nodes, windows of cells found by
    # searches on the keys, a second pass with wider margins, the single-target search) forced on: the same lists and
    # distances as the k-d tree -- uniform cloud, graded ones, targets beyond the sources' box, k above the lane kernel's
    # 20 (those queries take the level-0 grid)
    monkeypatch.setenv("MM_KNN_TREE", "1")
    rng = np.random.default_rng(29)
    src = rng.uniform(size=(150_000, 3)) ** power
    q = np.concatenate([rng.uniform(size=(40_000, 3)) ** power, rng.uniform(-0.2, 1.2, size=(5_000, 3)), src[:500]])
    ref, dref = O.knn_ckdtree(src, q, k, workers=-1)
    tree = ctx.knn_build(src)
    for _ in range(2):
        idx, dist = tree.query(q, k, want_dist=True)
        assert np.array_equal(idx.numpy().reshape(len(q), k), ref.reshape(len(q), k))
        np.testing.assert_allclose(dist.numpy().reshape(len(q), k), dref.reshape(len(q), k), rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_knn_tree_two_densities_and_a_cluster(ctx, monkeypatch):
    # a refined region (27x the density), and a cluster of 130 k sources inside one finest cell's reach: nodes that stay
    # too full however deep the tree goes -- their targets must come out of the single-target search
    monkeypatch.setenv("MM_KNN_TREE", "1")
    rng = np.random.default_rng(31)
    src = np.concatenate([rng.uniform(size=(120_000, 3)), 0.35 + 0.2 * rng.uniform(size=(120_000, 3)),
                          0.52 + 1e-6 * rng.normal(size=(131_000, 3))])
    q = np.concatenate([rng.uniform(size=(30_000, 3)), 0.35 + 0.2 * rng.uniform(size=(30_000, 3)),
                        0.52 + 1e-5 * rng.normal(size=(3_000, 3))])
    for k in (8, 20):
        assert np.array_equal(ctx.knn_build(src).query(q, k).numpy(), O.knn_ckdtree(src, q, k, workers=-1)[0])


@pytest.mark.gpu
def test_knn_tree_edge_cases(ctx, monkeypatch):
    # the smallest cloud the tree takes (4096 sources: a 2 M-entry search table over a handful of leaves), every target
    # outside the sources' box (all keys clamped into boundary cells), k = 1 and k = 20, sources on a plane of the cube
    # (two axes of full extent, one of 1e-9: nodes that are all slabs) and sixty-fold duplicates of one point
    monkeypatch.setenv("MM_KNN_TREE", "1")
    rng = np.random.default_rng(37)
    src = rng.uniform(size=(4096, 3))
    far = np.concatenate([rng.uniform(1.5, 3.0, size=(3000, 3)), rng.uniform(-2.0, -0.5, size=(3000, 3))])
    for k in (1, 20):
        assert np.array_equal(ctx.knn_build(src).query(far, k).numpy().reshape(len(far), k),
                              O.knn_ckdtree(src, far, k, workers=-1)[0].reshape(len(far), k))
    flat = rng.uniform(size=(60_000, 3)) * np.array([1.0, 1.0, 1e-9])
    q = rng.uniform(size=(20_000, 3)) * np.array([1.0, 1.0, 1e-9])
    assert np.array_equal(ctx.knn_build(flat).query(q, 8).numpy(), O.knn_ckdtree(flat, q, 8, workers=-1)[0])
    dup = np.concatenate([rng.uniform(size=(20_000, 3)), np.repeat(rng.uniform(size=(500, 3)), 60, axis=0)])
    q = rng.uniform(size=(10_000, 3))
    idx, dist = ctx.knn_build(dup).query(q, 8, want_dist=True)
    ref_i, ref_d = O.knn_ckdtree(dup, q, 8, workers=-1)
    np.testing.assert_allclose(dist.numpy().reshape(len(q), 8), ref_d.reshape(len(q), 8), rtol=1e-12, atol=0)
    far_from_dups = (ref_d.reshape(len(q), 8)[:, -1] < np.inf)   # ids: equal where no tie among duplicates decides
    same = (idx.numpy().reshape(len(q), 8) == ref_i.reshape(len(q), 8)).all(axis=1)
    assert same.mean() > 0.5 and far_from_dups.all()


@pytest.mark.gpu
@pytest.mark.parametrize("power", [1.5, 2.2])
def test_knn_tree_serves_the_fused_pipeline_on_a_graded_mesh(ctx, power, monkeypatch):
    # the whole hex8 path over the tree: lazily evaluated lists of 8 in Morton order (the locate stage walks the targets in
    # that order), the full lists of the targets that exhaust them through the tree again -- ids, weights, values and the
    # failed count equal to the oracle's, as with the stack of density levels
    monkeypatch.setenv("MM_KNN_TREE", "1")
    pa, ca = synth.hex_mesh(61, seed=1, jitter=0.1)
    pb, _ = synth.hex_mesh(75, seed=7, jitter=0.1)
    pa, pb = pa ** power, pb ** power
    fields = synth.vector_field(pa)[:2]
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
    vals_o = O.gather(fields, enc_o, w_o)
    for lazy in (True, False):
        ctx.set_lazy_lists(lazy)
        try:
            vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
        finally:
            ctx.set_lazy_lists(True)
        assert nf == nf_o
        assert np.array_equal(enc.numpy(), enc_o) and np.array_equal(w.numpy(), w_o)
        assert np.array_equal(vals.numpy(), vals_o)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [4, 8])
def test_knn_lane_kernel_next_to_a_cell_with_more_than_65535_sources(ctx, k):
    # a clustered cloud: one level-0 cell holds 131 k sources.  The lane kernel (level 0 of a multi-level grid for
    # k <= 8) packs two running cell counts into one 32-bit word; found by tools/fuzz_knn.py (seed 501, case 2017):
    # the count wrapped, the tile looked as if it fitted, and targets near the cluster got far-away neighbours
    rng = np.random.default_rng(17)
    src = np.concatenate([rng.uniform(size=(40_000, 3)), 0.52 + 1e-3 * rng.normal(size=(2 * 65536 + 300, 3))])   # = 300 mod 2^16
    q = np.concatenate([rng.uniform(-0.1, 1.1, size=(70_000, 3)), 0.52 + 0.05 * rng.normal(size=(20_000, 3))])
    idx = ctx.knn_build(src).query(q, k).numpy()
    assert np.array_equal(idx, O.knn_ckdtree(src, q, k, workers=-1)[0])


_LANE_WINDOW_CHECK = r"""
import sys
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
from oracle import oracle as O
ctx = Context(0)
rng = np.random.default_rng(33)
# (a) mesh-like: centroids of a jittered hex mesh, targets = nodes of another one + random points; (b) a random cloud
pa, ca = synth.hex_mesh(61, seed=1)
cen = O.centroid(ca, pa)
pb, _ = synth.hex_mesh(70, seed=7)
for src, q in ((cen, np.concatenate([pb, rng.uniform(-0.05, 1.05, size=(150_000, 3))])),
               (rng.uniform(size=(300_000, 3)), rng.uniform(-0.02, 1.02, size=(600_000, 3)))):
    tree = ctx.knn_build(src)
    for k in (1, 4, 8):
        idx, dist = tree.query(q, k, want_dist=True)
        ref, refd = O.knn_ckdtree(src, q, k, workers=-1)
        assert np.array_equal(idx.numpy(), ref), (k, int((idx.numpy() != ref).any(axis=1).sum()))
        diff = src[ref] - q[:, None, :]
        assert np.array_equal(dist.numpy(), np.sqrt((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]))
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"MM_KNN_LANE_T": "6", "MM_KNN_LANE_W": "1"},    # nearly every round is widened
                                   {"MM_KNN_LANE_T": "6", "MM_KNN_LANE_W": "3"},
                                   {"MM_KNN_LANE_T": "3", "MM_KNN_LANE_W": "2", "MM_KNN_LANE_Z": "12"},
                                   {"MM_KNN_LANE_T": "1", "MM_KNN_LANE_W": "1"}])  # whole cell layers, as in round 2
def test_knn_lane_kernel_windows_of_thin_layers_and_their_widening(knobs):
    # the lane kernel first scans 2 W + 1 thin layers of the tile and, when a round cannot be certified, the
    # entries a full cell layer adds on either side.  The knobs are read once per process, so every setting runs
    # in a child process: forced to the lane kernel, narrow windows make the widening path the common one.
    import os
    import subprocess
    import sys

    env = dict(os.environ, MM_KNN_KERNEL="lane", **knobs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _LANE_WINDOW_CHECK], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.gpu
def test_knn_list_mode_one_wave_per_target(ctx, monkeypatch):
    # the kernel that serves the targets the fast kernels hand over (and the locate stage's lazily fetched full
    # lists), here given EVERY target: rings, pruning, the wave-wide merge, ties by index, pads, density levels
    monkeypatch.setenv("MM_KNN_FORCE_LIST", "1")
    rng = np.random.default_rng(21)
    src = rng.uniform(size=(60_000, 3))
    q = rng.uniform(-0.3, 1.3, size=(9_000, 3))
    for k in (1, 8, 20, 32):
        idx, dist = ctx.knn_build(src).query(q, k, want_dist=True)
        ref, _ = O.knn_ckdtree(src, q, k, workers=-1)
        diff = src[ref] - q[:, None, :]
        refd = np.sqrt((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2])
        assert np.array_equal(idx.numpy(), ref) and np.array_equal(dist.numpy(), refd)
    pa, ca = synth.hex_mesh(9, jitter=0.0)                       # exact ties
    cen = O.centroid(ca, pa)
    assert np.array_equal(ctx.knn_build(cen).query(pa, 20).numpy(), O.knn_brute(cen, pa, 20))
    graded = rng.uniform(size=(80_000, 3)) ** 3.0                # several density levels
    assert np.array_equal(ctx.knn_build(graded).query(q[:3000], 20).numpy(), O.knn_ckdtree(graded, q[:3000], 20)[0])
    flat = rng.uniform(size=(5_000, 2))                          # 2-D
    q2 = rng.uniform(-0.2, 1.2, size=(800, 2))
    assert np.array_equal(ctx.knn_build(flat).query(q2, 16).numpy(), O.knn_ckdtree(flat, q2, 16)[0])
    few = rng.uniform(size=(5, 3))                               # fewer sources than k: padded
    idx, dist = ctx.knn_build(few).query(q[:50], 8, want_dist=True)
    assert np.array_equal(idx.numpy()[:, :5], O.knn_brute(few, q[:50], 5)) and np.all(idx.numpy()[:, 5:] == 5)
    assert np.all(np.isinf(dist.numpy()[:, 5:]))


@pytest.mark.gpu
def test_bad_arguments_are_refused_with_a_code_and_a_message(ctx):
    # the reference has no error channel at all; ours: negative MM_ERR_* and a
    # thread-local message, nothing launched, and the context stays usable
    lib, h = ctx.lib, ctx.handle
    pa, ca = synth.hex_mesh(5, seed=1)
    d_nodes, d_conn, d_pts = ctx.to_device(pa), ctx.to_device(ca), ctx.to_device(pa[:10])
    d_f = ctx.to_device(pa[:, 0].copy())
    out = ctx.empty((10, 1), np.float64)
    MM_ERR_ARG = -1
    cases = [
        lambda: lib.mm_interpolate_hex8(h, d_nodes.ptr, len(pa), d_conn.ptr, len(ca), d_pts.ptr, 10, d_f.ptr, 1, 0,
                                        out.ptr, None, None),                      # k = 0
        lambda: lib.mm_interpolate_hex8(h, d_nodes.ptr, len(pa), d_conn.ptr, len(ca), d_pts.ptr, 10, d_f.ptr, 1, 65,
                                        out.ptr, None, None),                      # k > MM_KNN_MAX_K
        lambda: lib.mm_interpolate_hex8(h, None, len(pa), d_conn.ptr, len(ca), d_pts.ptr, 10, d_f.ptr, 1, 20,
                                        out.ptr, None, None),                      # null nodes
        lambda: lib.mm_interpolate_hex8(h, d_nodes.ptr, len(pa), d_conn.ptr, len(ca), d_pts.ptr, -1, d_f.ptr, 1, 20,
                                        out.ptr, None, None),                      # negative size
        lambda: lib.mm_centroid(h, 4, len(ca), 8, d_conn.ptr, d_nodes.ptr, out.ptr),                 # ndim = 4
        lambda: lib.mm_knn_query(h, None, d_pts.ptr, 10, 5, out.ptr, None),                           # null index
        lambda: lib.mm_gather(h, d_f.ptr, len(pa), 1, None, None, 10, 8, out.ptr, 1),                 # null operator
        lambda: lib.mm_gather(h, d_f.ptr, len(pa), 1, d_conn.ptr, d_nodes.ptr, 10, 129, out.ptr, 1),  # P too large
        lambda: lib.mm_locate_gll(h, 3, 3, 5, 10, d_conn.ptr, d_nodes.ptr, 1, d_pts.ptr, 1.05, 0, out.ptr, out.ptr),
        lambda: lib.mm_interpolate_gll(h, 4, 3, d_nodes.ptr, 1, d_pts.ptr, 10, d_f.ptr, 1, 20, 1.05, 0, out.ptr,
                                       out.ptr, None),                             # operator outputs: both or neither
        lambda: lib.mm_unique_points(h, d_pts.ptr, 10, 4, out.ptr, out.ptr),                          # dim = 4
    ]
    for call in cases:
        assert call() == MM_ERR_ARG
        assert len(lib.mm_last_error()) > 0
    # still alive
    vals, nf = ctx.interpolate_hex8(pa, ca, pa[:10], pa[:, 0].copy(), nelem_to_search=20)
    assert nf == 0 and np.abs(vals.numpy()[:, 0] - pa[:10, 0]).max() < 1e-7    # Newton tolerance 1e-8 x element size


@pytest.mark.gpu
def test_stage_timers_levels(ctx):
    # mm_set_profiling: 1 times every stage of a call, 2 only the two dominant kernels (what bench.py runs its timed
    # steps with: a timed stage costs the stream two events), 0 nothing; the results do not depend on it
    pa, ca = synth.hex_mesh(30, seed=1, jitter=0.2)
    pb, _ = synth.hex_mesh(31, seed=7, jitter=0.2)
    fields = synth.vector_field(pa)[:1]
    got = {}
    try:
        for level in (True, 2, False):
            ctx.set_profiling(level)
            vals, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
            t = ctx.last_timings()
            got[level] = vals.numpy()
            timed = {s for s, v in t.items() if v > 0.0}
            if level is True:
                assert {"centroid", "knn_build", "knn_query", "locate", "locate_pass0"} <= timed
            elif level == 2:
                assert timed <= {"knn_cell", "locate_pass0"} and "locate_pass0" in timed
            else:
                assert not timed
    finally:
        ctx.set_profiling(False)
    assert np.array_equal(got[True], got[2]) and np.array_equal(got[True], got[False])


# ------------------------------------------------------------------------------- resident source
def test_resident_source_gives_the_fused_path_bit_for_bit(ctx, golden):
    # mm_source_create builds centroids + search grid once (the reference builds its cKDTree once and queries it for
    # every GLL point / time step, scripts/cli.py:141-195); mm_interpolate_hex8_on is the fused path without those two
    # stages.  Same bits as mm_interpolate_hex8 and as the reference fixtures, call after call, lazy and eager lists.
    d = golden("hex8_hard_k20")
    src = ctx.source(d["points_a"], d["conn_a"])
    try:
        for lazy in (True, False, True):
            ctx.set_lazy_lists(lazy)
            try:
                vals, enc, w, nf = src.interpolate(d["points_b"], d["fields"], nelem_to_search=int(d["k"]), want_operator=True)
                vals2, nf2 = src.interpolate(d["points_b"], d["fields"], nelem_to_search=int(d["k"]))
            finally:
                ctx.set_lazy_lists(True)
            assert nf == nf2 == int(d["nfailed"])
            assert np.array_equal(enc.numpy(), d["enc"]) and np.array_equal(w.numpy(), d["w"])
            assert vals.numpy().tobytes() == np.ascontiguousarray(d["values"]).tobytes() == vals2.numpy().tobytes()
    finally:
        src.free()
    # a larger mesh, many target sets against one source (and the fused path with its own rebuild in between)
    pa, ca = synth.hex_mesh(40, seed=1)
    fields = synth.vector_field(pa)[:2]
    src = ctx.source(pa, ca)
    try:
        for seed in (7, 8, 9):
            pb, _ = synth.hex_mesh(37 + seed, seed=seed)
            a, nfa = src.interpolate(pb, fields)
            b, nfb = ctx.interpolate_hex8(pa, ca, pb, fields)
            assert nfa == nfb == 0 and np.array_equal(a.numpy(), b.numpy())
    finally:
        src.free()


def test_resident_source_on_a_graded_mesh(ctx):
    # a graded source mesh kept resident: its index is the density-adaptive tree (as in the per-call build); the tree's
    # arrays belong to the handle, queries of several target sets interleave with fused calls that build their own
    pa, ca = synth.hex_mesh(61, seed=1, jitter=0.1)
    pa = pa ** 1.8
    fields = synth.vector_field(pa)[:2]
    src = ctx.source(pa, ca)
    try:
        for seed, n in ((7, 75), (8, 50), (9, 64)):
            pb, _ = synth.hex_mesh(n, seed=seed, jitter=0.1)
            pb = pb ** 1.8
            a, nfa = src.interpolate(pb, fields, nelem_to_search=20)
            b, nfb = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
            assert nfa == nfb and np.array_equal(a.numpy(), b.numpy())
            nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
            enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
            assert nf_o == nfa and np.array_equal(a.numpy(), O.gather(fields, enc_o, w_o))
    finally:
        src.free()


# ------------------------------------------------------------------------------- targets that fill part of the grid
def test_knn_targets_in_a_slab_choose_their_kernel_by_occupied_strips(ctx):
    # One rank's share of a sharded target set is a slab: the full problem's density inside, nothing outside.  The average
    # density test (npts >= 2 * ncells) fails for it; the query then counts the lane kernel's work items on the device once
    # (first call: a readback; later calls of the same sizes: the kept verdict) and takes the lane kernel when the occupied
    # strips are well filled.  Both calls, dense slab and thin uniform cloud, k = 8 and 20: indices and distances equal
    # cKDTree's bit for bit.
    rng = np.random.default_rng(12)
    src = rng.uniform(size=(400_000, 3))                       # ~50 k cells
    tree = ctx.knn_build(src)
    slab = rng.uniform(size=(60_000, 3)) * np.array([0.12, 1.0, 1.0]) + np.array([0.3, 0.0, 0.0])   # 1/8 of the box, 10 per cell
    thin = rng.uniform(size=(40_000, 3))                       # < 1 per cell everywhere
    for pts in (slab, thin):
        for k in (8, 20):
            want_i, want_d = O.knn_ckdtree(src, pts, k, workers=-1)
            for _ in range(2):                                  # probe, then the remembered verdict
                got_i, got_d = tree.query(pts, k, want_dist=True)
                assert np.array_equal(got_i.numpy(), want_i) and np.array_equal(got_d.numpy(), want_d)


def test_a_strong_scaling_shard_equals_its_rows_of_the_whole_run(ctx):
    # rank 3 of 8 of a strong-scaling run (bench.py --as-rank 3/8): rows shard_bounds(N, 8, 3) of the target mesh, an
    # x-slab of the domain, through the fused pipeline twice (kernel probe, then the kept verdict) -- bit-equal to the same
    # rows of the run over the whole target set, in both arithmetics' ids
    from multimesh_amd.distributed import shard_bounds

    pa, ca = synth.hex_mesh(80, seed=1)
    pb, _ = synth.hex_mesh(80, seed=7)
    fields = synth.vector_field(pa)[:2]
    whole, enc_w, w_w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, want_operator=True)
    lo, hi = shard_bounds(len(pb), 8, 3)
    for _ in range(2):
        part, enc_p, w_p, nfp = ctx.interpolate_hex8(pa, ca, pb[lo:hi], fields, want_operator=True)
        assert nf == nfp == 0
        assert np.array_equal(part.numpy(), whole.numpy()[lo:hi])
        assert np.array_equal(enc_p.numpy(), enc_w.numpy()[lo:hi]) and np.array_equal(w_p.numpy(), w_w.numpy()[lo:hi])
