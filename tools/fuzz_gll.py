"""Randomised check of mm_locate_gll (tolerance / snap loop) and mm_locate_gll_bbox (bounding-box loop)
plus mm_gather_elem and the fused mm_interpolate_gll against the oracle's restatement: orders 1, 2, 4 in 2-D and 3-D, distorted meshes,
targets inside, on and outside the hull, random k / tolerance / snapping.  Bit-equal element ids,
coefficients and gathered values are required.  Not part of the test suite."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
from oracle import oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = Context(0)
t0 = time.time()
for case in range(ncases):
    order = int(rng.choice([1, 2, 4]))
    dim = int(rng.choice([2, 3]))
    n = int(rng.integers(3, 9 if dim == 3 else 24))
    jitter = float(rng.uniform(0.0, 0.3))
    src = synth.gll_mesh(n, order, seed=int(rng.integers(1, 1 << 30)), dim=dim, jitter=jitter)
    if rng.random() < 0.5:
        src = src * rng.uniform(0.3, 4.0, size=dim) + rng.uniform(-50, 50, size=dim)
        src[..., 0] += rng.uniform(-0.4, 0.4) * src[..., 1]
    lo, hi = src.reshape(-1, dim).min(axis=0), src.reshape(-1, dim).max(axis=0)
    npts = int(rng.integers(1, 6000))
    margin = rng.choice([0.0, 0.03, 0.4])
    pts = rng.uniform(lo - margin * (hi - lo), hi + margin * (hi - lo), size=(npts, dim))
    k = int(rng.choice([1, 2, 5, 9, 12, 20, 27]))
    k = min(k, len(src))
    cen = src.mean(axis=1) + rng.normal(scale=1e-9, size=(len(src), dim))     # general position
    nn, _ = O.knn_ckdtree(cen, pts, k, workers=-1)
    nn = nn.reshape(npts, k)
    tol = float(rng.choice([1.0, 1.03, 1.05, 1.2]))
    snap = bool(rng.random() < 0.5)
    ncomp = int(rng.choice([1, 2]))
    fields = rng.normal(size=(ncomp,) + src.shape[:2])
    e_g, c_g, m_g = ctx.locate_gll(order, nn, src, pts, tol, snap)
    e_o, c_o, m_o = O.locate_gll(order, nn, src, pts, tol, snap)
    good = m_g == m_o and np.array_equal(e_g.numpy(), e_o) and np.array_equal(c_g.numpy(), c_o)
    v_g = ctx.gather_elem(fields, e_g, c_g).numpy()
    good = good and v_g.tobytes() == O.gather_elem(fields, e_o, c_o).tobytes()
    e_b, c_b, h_b = ctx.locate_gll_bbox(order, nn, src, pts)
    e_bo, c_bo, h_bo = O.locate_gll_v1(order, nn, src, pts)
    good_b = h_b == h_bo and np.array_equal(e_b.numpy(), e_bo) and np.array_equal(c_b.numpy(), c_bo)
    # the fused entry (own centroids and kNN; lists lazy or eager; with or without the operator) -- on
    # meshes in general position only: tied centroid distances are ordered differently by cKDTree
    good_f = True
    if jitter > 0.02:
        nn_f = O.knn_ckdtree(src.mean(axis=1), pts, k, workers=-1)[0].reshape(npts, k)
        e_f, c_f, m_f = O.locate_gll(order, nn_f, src, pts, tol, snap)
        v_f = O.gather_elem(fields, e_f, c_f)
        lazy, want_op = bool(rng.random() < 0.7), bool(rng.random() < 0.3)
        ctx.set_lazy_lists(lazy)
        r = ctx.interpolate_gll(order, src, pts, fields, nelem_to_search=k, tolerance=tol, snap_to_nearest=snap,
                                want_operator=want_op)
        ctx.set_lazy_lists(True)
        good_f = r[-1] == m_f and r[0].numpy().tobytes() == v_f.tobytes()
        if want_op:
            good_f = good_f and np.array_equal(r[1].numpy(), e_f) and np.array_equal(r[2].numpy(), c_f)
    good = good and good_f
    print(f"case {case:3d} order={order} dim={dim} n={n:2d} N={npts:5d} k={k:2d} tol={tol} snap={int(snap)} margin={margin} "
          f"missing={m_g:5d} hard={h_b:4d} -> {'ok' if good and good_b else 'MISMATCH'}", flush=True)
    if not (good and good_b):
        print("  tolerance loop + fused ok:", good, "(fused:", good_f, ") bbox loop ok:", good_b)
        sys.exit(1)
print(f"{ncases} cases ok in {time.time() - t0:.0f} s")
