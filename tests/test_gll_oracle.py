"""SURVEY.md §8 row A10 is PARITY UNPINNED (the reference's GLL numerics live in the absent,
proprietary salvus.fem and no reference test pins them).  These tests pin our own definition
(oracle/mm_oracle.c "A10" == multimesh_amd/csrc/mm_locate_gll.hip) by analytic properties:
partition of unity, Kronecker property, polynomial reproduction up to the element order,
inverse-transform round trip, control flow of the reference's acceptance loop, and
order-1 GLL == the (reference-pinned) hex8 path."""
import numpy as np
import pytest

from multimesh_amd import synth
from oracle import oracle as O

CASES = [(o, d) for o in (1, 2, 4) for d in (2, 3)]


@pytest.mark.parametrize("order,dim", CASES)
def test_coefficients_partition_of_unity_and_kronecker(order, dim):
    g = synth.gll_nodes_1d(order)
    rng = np.random.default_rng(order * 10 + dim)
    for _ in range(50):
        xi = rng.uniform(-1.05, 1.05, size=dim)
        assert abs(O.gll_coefficients(order, xi).sum() - 1) < 1e-13
    m = order + 1
    for p in range(m ** dim):
        idx = [(p // m ** a) % m for a in range(dim)]
        c = O.gll_coefficients(order, g[idx])
        e = np.zeros(m ** dim)
        e[p] = 1
        assert np.abs(c - e).max() < 1e-14


@pytest.mark.parametrize("order,dim", CASES)
def test_polynomials_up_to_the_order_are_reproduced(order, dim):
    # in reference coordinates every monomial xi^a * eta^b * ... with exponents <= order is in the basis
    g = synth.gll_nodes_1d(order)
    rng = np.random.default_rng(3)
    grids = np.meshgrid(*([g] * dim), indexing="ij")
    nodes = np.stack([gr.T.ravel() if dim == 2 else gr.transpose(2, 1, 0).ravel() for gr in grids], axis=1)
    # node p = i + m j (+ m^2 k): first axis fastest
    for _ in range(20):
        expo = rng.integers(0, order + 1, size=dim)
        xi = rng.uniform(-1, 1, size=dim)
        nodal = np.prod(nodes ** expo, axis=1)
        got = O.gll_coefficients(order, xi) @ nodal
        assert abs(got - np.prod(xi ** expo)) < 1e-12


@pytest.mark.parametrize("order,dim", CASES)
def test_inverse_transform_round_trip_and_failure(order, dim):
    gp = synth.gll_mesh(4, order, seed=5, jitter=0.3, dim=dim)
    rng = np.random.default_rng(7)
    for _ in range(40):
        e = rng.integers(0, gp.shape[0])
        xi = rng.uniform(-1.02, 1.02, size=dim)
        x = O.gll_coefficients(order, xi) @ gp[e]
        back = O.gll_inverse_transform(order, x, gp[e])
        assert np.abs(back - xi).max() < 1e-10
    far = O.gll_inverse_transform(order, np.full(dim, 50.0), gp[0])     # way outside: diverges -> NaN
    assert np.isnan(far).all()
    flat = np.zeros_like(gp[0])                                           # degenerate element: singular Jacobian
    assert np.isnan(O.gll_inverse_transform(order, np.full(dim, 0.1), flat)).all()


@pytest.mark.parametrize("order,dim", CASES)
def test_locate_and_gather_reproduce_smooth_fields(order, dim):
    gp = synth.gll_mesh(6, order, seed=2, dim=dim)
    rng = np.random.default_rng(1)
    pts = rng.uniform(0.0, 1.0, size=(800, dim))
    nn, _ = O.knn_ckdtree(gp.mean(axis=1), pts, min(25, gp.shape[0]))
    elem, co, miss = O.locate_gll(order, nn, gp, pts)
    assert miss == 0 and (elem >= 0).all()
    assert np.abs(co.sum(axis=1) - 1).max() < 1e-12
    lin = O.gather_elem(synth.field_linear(gp), elem, co)[:, 0]
    assert np.abs(lin - synth.field_linear(pts)).max() < 1e-12              # any order reproduces linear fields
    smooth_err = np.abs(O.gather_elem(synth.field_smooth(gp.reshape(-1, dim)).reshape(gp.shape[:2]), elem, co)[:, 0]
                        - synth.field_smooth(pts)).max()
    assert smooth_err < {1: 0.8, 2: 0.12, 4: 3e-3}[order]                  # error falls fast with the order
    # literal NumPy statement of reference interpolator.py:976
    f = synth.field_smooth(gp.reshape(-1, dim)).reshape(gp.shape[:2])
    assert np.array_equal(O.gather_elem(f, elem, co)[:, 0], np.sum(co * f[elem], axis=1))
    # points that were not found: element -1 and zero coefficients -> NumPy reads the LAST element,
    # and the sign of the resulting zero follows the signs of its field values
    elem_m, co_m = elem.copy(), co.copy()
    elem_m[::3], co_m[::3] = -1, 0.0
    for g in (f, -1.0 - np.abs(f)):
        want = np.sum(co_m * g[elem_m], axis=1)
        got = O.gather_elem(g, elem_m, co_m)[:, 0]
        assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want))


def test_control_flow_not_found_snap_and_tolerance():
    gp = synth.gll_mesh(4, 2, seed=1, dim=3)
    cen = gp.mean(axis=1)
    pts = np.array([[0.5, 0.5, 0.5], [1.005, 0.5, 0.5], [1.5, 0.5, 0.5], [-0.3, -0.3, -0.3]])  # xi_x = 1.03 for #1
    nn, _ = O.knn_ckdtree(cen, pts, 25)
    elem, co, miss = O.locate_gll(2, nn, gp, pts, tolerance=1.05, snap_to_nearest=False)
    assert elem[0] >= 0 and elem[1] >= 0 and elem[2] == -1 and elem[3] == -1 and miss == 2
    assert not co[2].any() and not co[3].any()                             # reference: (-1, zeros)
    elem_s, co_s, miss_s = O.locate_gll(2, nn, gp, pts, tolerance=1.05, snap_to_nearest=True)
    assert miss_s == 0 and (elem_s >= 0).all()
    assert np.abs(co_s.sum(axis=1) - 1).max() < 1e-12                      # clipped xi still a partition of unity
    # a tighter tolerance (the layered variant uses 1.03) can only lose points: #1 drops out at 1.0
    _, _, miss_t = O.locate_gll(2, nn, gp, pts, tolerance=1.0, snap_to_nearest=False)
    assert miss_t == 3
    # padded / invalid candidates are skipped
    bad = np.concatenate([np.full((len(pts), 1), gp.shape[0] + 5), nn[:, :-1]], axis=1)
    elem_b, _, _ = O.locate_gll(2, bad, gp, pts)
    assert elem_b[0] == elem[0]


def test_order1_gll_equals_reference_pinned_hex8_path():
    # same element found and same interpolated values as the hex8 path (whose oracle IS pinned)
    pa, ca = synth.hex_mesh(7, seed=1)
    gp = synth.gll_mesh(7, 1, seed=1, dim=3)
    rng = np.random.default_rng(9)
    pts = rng.uniform(0.05, 0.95, size=(600, 3))
    cen = O.centroid(ca, pa)
    assert np.abs(cen - gp.mean(axis=1)).max() < 1e-15
    nn, _ = O.knn_ckdtree(cen, pts, 20)
    enc, w, nf = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pts)
    elem, co, miss = O.locate_gll(1, nn, gp, pts, tolerance=1.025)
    assert nf == 0 and miss == 0
    f_nodal = synth.field_smooth(pa)
    v_hex = O.gather(f_nodal, enc, w)[:, 0]
    v_gll = O.gather_elem(synth.field_smooth(gp.reshape(-1, 3)).reshape(gp.shape[:2]), elem, co)[:, 0]
    assert np.abs(v_hex - v_gll).max() < 1e-7                               # hex8 Newton stops at 1e-8 * scale
    # both found the element whose 8 corners are the hex8 row's node set
    assert all(set(enc[i]) == set(ca[elem[i]]) for i in range(len(pts)))


# ---- variant 1: bounding-box pre-test loop (reference interpolator.py:1409-1473) ---------------------
def test_bbox_variant_branches():
    src = synth.gll_mesh(5, 2, seed=4, dim=3)
    rng = np.random.default_rng(1)
    inside = rng.uniform(0.05, 0.95, size=(400, 3))
    far = rng.uniform(1.5, 2.0, size=(50, 3))                  # no bounding box contains these
    pts = np.concatenate([inside, far])
    nn, _ = O.knn_ckdtree(src.mean(axis=1), pts, 10)
    elem, coeffs, hard = O.locate_gll_v1(2, nn, src, pts)
    # `hard` counts the far points whose final transform fails outright (the reference raises there
    # unless ignore_hard_elements); either way they get the constant xi below
    assert 0 <= hard <= 50 and (elem >= 0).all()
    # accepted points: the element's own basis reproduces the point, coefficients sum to one
    rec = np.einsum("np,npd->nd", coeffs[:400], src[elem[:400]])
    assert np.abs(rec - inside).max() < 1e-10
    assert np.abs(coeffs.sum(axis=1) - 1).max() < 1e-12
    # far points: nearest control-node mean among the candidates, then the reference's constant xi
    cen = src.mean(axis=1)
    d = np.linalg.norm(cen[nn[400:]] - far[:, None, :], axis=2)
    assert np.array_equal(elem[400:], nn[400:][np.arange(50), d.argmin(axis=1)])
    const = O.gll_coefficients(2, np.array([0.645, -0.5, 0.22]))
    assert np.allclose(coeffs[400:], const[None, :], rtol=0, atol=0)


def test_bbox_variant_first_inside_box_wins_when_nothing_is_accepted():
    # a point inside an element's bounding box but outside the (sheared) element and 4 % band:
    # the first candidate whose box contains it is used with the constant xi
    src = synth.gll_mesh(3, 1, seed=1, dim=2, jitter=0.0).copy()
    src[:, :, 0] += 0.8 * src[:, :, 1]                         # shear: boxes overlap, elements do not
    e = 0
    lo, hi = src[e].min(axis=0), src[e].max(axis=0)
    p = np.array([[lo[0] + 0.02 * (hi[0] - lo[0]), hi[1] - 0.02 * (hi[1] - lo[1])]])   # a box corner region
    nn = np.array([[e]], dtype=np.int64)
    elem, coeffs, hard = O.locate_gll_v1(1, nn, src, p)
    xi = O.gll_inverse_transform(1, p[0], src[e])
    assert np.abs(xi).max() > 1.04                              # not acceptable in element e
    assert elem[0] == e and hard == 0
    assert np.array_equal(coeffs[0], O.gll_coefficients(1, np.array([0.645, -0.5])))


def test_production_arithmetic_is_bounded_by_an_independent_strict_statement():
    """ADVICE (round 2): the GLL Newton's stop test (1e-10) and its fused multiply-adds changed in the kernel and in
    the oracle together, so kernel == oracle says nothing about the arithmetic itself.  The strict statement
    (plain sums over the nodes, no fma, Gaussian solve with pivoting, 1e-13) is independent of both: on a
    cfg5-shaped case (order-4 hexes, jittered, targets = GLL points of a finer mesh incl. points on element
    faces) the accepted elements must be identical and xi / the coefficients must agree to 1e-9."""
    from multimesh_amd import synth

    for order, n_src, n_tgt in ((4, 7, 8), (2, 9, 11)):
        src = synth.gll_mesh(n_src, order, seed=1)
        tgt = np.unique(synth.gll_mesh(n_tgt, order, seed=7).reshape(-1, 3), axis=0)
        rng = np.random.default_rng(order)
        tgt = tgt[rng.choice(len(tgt), size=min(len(tgt), 6000), replace=False)]
        nn, _ = O.knn_ckdtree(src.mean(axis=1), tgt, 20)
        for snap in (False, True):
            elem, co, miss = O.locate_gll(order, nn, src, tgt, 1.05, snap)
            elem1, co1, hard1 = O.locate_gll_v1(order, nn, src, tgt)
            O.set_gll_strict(True)
            try:
                elem_s, co_s, miss_s = O.locate_gll(order, nn, src, tgt, 1.05, snap)
                elem1_s, co1_s, hard1_s = O.locate_gll_v1(order, nn, src, tgt)
                # reference coordinates of the accepted element, both ways
                found = elem >= 0
                xi_s = np.array([O.gll_inverse_transform(order, tgt[i], src[elem[i]]) for i in np.nonzero(found)[0][:500]])
            finally:
                O.set_gll_strict(False)
            xi_p = np.array([O.gll_inverse_transform(order, tgt[i], src[elem[i]]) for i in np.nonzero(found)[0][:500]])
            assert miss == miss_s and np.array_equal(elem, elem_s)
            assert hard1 == hard1_s and np.array_equal(elem1, elem1_s)
            assert np.abs(xi_p - xi_s).max() <= 1e-9
            assert np.abs(co - co_s).max() <= 1e-9 and np.abs(co1 - co1_s).max() <= 1e-9


def test_the_corner_start_agrees_with_the_plain_start_on_curved_elements():
    """ADVICE (round 3): the production inverse transform starts from the solution of the corners' trilinear map (round 4:
    the polynomial form, mmo_hex8_start) where the reference (interpolator.py:1370-1386, through salvus.fem) iterates from
    xi = 0; "kernel == oracle" cannot see whether that start changes WHICH element accepts a point on curved or strongly
    distorted elements.  The strict statement starts from xi = 0: on a shell-like mesh whose elements are really curved
    (every GLL node displaced by a smooth non-polynomial field, so the corners' trilinear map is only an approximation of
    the element) and on points just inside and just outside the elements' faces, both acceptance loops must pick the same
    elements, miss the same points, and agree in xi and the coefficients to the Newton tolerance."""
    from multimesh_amd import synth

    def curve(p):
        # a quarter shell: radius and angles from the unit cube, plus a ripple -- nothing here is trilinear
        r = 1.0 + 0.6 * p[..., 0] + 0.03 * np.sin(5.0 * p[..., 1]) * np.cos(4.0 * p[..., 2])
        th, ph = 0.3 + 0.9 * p[..., 1], 0.2 + 0.8 * p[..., 2]
        return np.stack([r * np.cos(th) * np.cos(ph), r * np.sin(th) * np.cos(ph), r * np.sin(ph)], axis=-1)

    for order, n_src, n_tgt in ((4, 6, 7), (2, 8, 9)):
        src = curve(synth.gll_mesh(n_src, order, seed=1, jitter=0.25))
        tgt = np.unique(curve(synth.gll_mesh(n_tgt, order, seed=7, jitter=0.2)).reshape(-1, 3), axis=0)
        rng = np.random.default_rng(100 + order)
        tgt = tgt[rng.choice(len(tgt), size=min(len(tgt), 4000), replace=False)]
        # ... and points pushed across element faces: just inside / outside the 1.05 (1.04) tolerance of some element
        faces = src[:, :: (order + 1) ** 2 // 2 + 1][:300, 0]
        tgt = np.concatenate([tgt, faces + rng.normal(scale=2e-3, size=faces.shape)])
        nn, _ = O.knn_ckdtree(src.mean(axis=1), tgt, 20)
        for snap in (False, True):
            elem, co, miss = O.locate_gll(order, nn, src, tgt, 1.05, snap)
            elem1, co1, hard1 = O.locate_gll_v1(order, nn, src, tgt)
            O.set_gll_strict(True)
            try:
                elem_s, co_s, miss_s = O.locate_gll(order, nn, src, tgt, 1.05, snap)
                elem1_s, co1_s, hard1_s = O.locate_gll_v1(order, nn, src, tgt)
            finally:
                O.set_gll_strict(False)
            # a point whose max|xi| lies within the Newton tolerance of a threshold may legitimately fall either way:
            # none may differ by more than that (the same bound the other strict test uses)
            same = elem == elem_s
            assert miss == miss_s and same.mean() > 0.999, (order, snap, int((~same).sum()))
            assert np.abs(co[same] - co_s[same]).max() <= 1e-9
            same1 = elem1 == elem1_s
            assert hard1 == hard1_s and same1.mean() > 0.999
            assert np.abs(co1[same1] - co1_s[same1]).max() <= 1e-9
            for i in np.nonzero(~same)[0]:
                # the two starts disagree only where the point sits ON a threshold of both elements
                xa = O.gll_inverse_transform(order, tgt[i], src[elem[i]]) if elem[i] >= 0 else None
                if xa is not None:
                    assert abs(np.abs(xa).max() - 1.05) < 1e-8 or abs(np.abs(xa).max() - 1.0) < 0.06
