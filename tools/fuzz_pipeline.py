"""Randomised end-to-end check of mm_interpolate_hex8 against the oracle (cKDTree + C restatement +
NumPy-order gather): random mesh sizes, shears, anisotropy, k, component counts, target clouds that
lie inside, on and outside the hull, lazy and eager candidate lists.  Not part of the test suite (it
takes minutes); prints one line per case and exits non-zero on the first mismatch."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
from oracle import oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
ctx = Context(0)
t_start = time.time()
for case in range(ncases):
    n = int(rng.integers(4, 42))
    pa, ca = synth.hex_mesh(n, seed=int(rng.integers(1, 1 << 30)), jitter=float(rng.uniform(0.0, 0.3)) + 1e-3)
    pa = pa.copy()
    if rng.random() < 0.5:                                   # anisotropy / shear / offset
        pa *= rng.uniform(0.2, 5.0, size=3)
        pa[:, 0] += rng.uniform(-0.6, 0.6) * pa[:, 1]
        pa += rng.uniform(-1e3, 1e3, size=3)
    # general position (no exact kNN ties): a tiny random perturbation of every node
    pa += rng.normal(scale=1e-9 * np.ptp(pa, axis=0).max(), size=pa.shape)
    lo, hi = pa.min(axis=0), pa.max(axis=0)
    npts = int(rng.integers(1, 60_000))
    margin = rng.choice([0.0, 0.02, 0.3])
    pb = rng.uniform(lo - margin * (hi - lo), hi + margin * (hi - lo), size=(npts, 3))
    if rng.random() < 0.3:                                   # some targets exactly on mesh nodes
        take = rng.integers(0, len(pa), size=min(npts, 500))
        pb[: len(take)] = pa[take]
    k = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 20, 25, 32]))
    k = min(k, len(ca))
    ncomp = int(rng.choice([1, 1, 2, 3, 5]))
    fields = rng.normal(size=(ncomp, len(pa)))
    lazy = bool(rng.random() < 0.7)
    ctx.set_lazy_lists(lazy)
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=k, want_operator=True)
    vals2, nf2 = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=k)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, k, workers=-1)
    nn = nn.reshape(npts, k)
    enc_o, w_o, nf_o, status = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb, want_status=True)
    ok = status >= 0
    good = (nf == nf_o == nf2 and np.array_equal(enc.numpy()[ok], enc_o[ok]) and np.array_equal(w.numpy()[ok], w_o[ok])
            and not enc.numpy()[~ok].any() and not w.numpy()[~ok].any())
    enc_z, w_z = enc_o.copy(), w_o.copy()
    enc_z[~ok] = 0
    w_z[~ok] = 0
    ref_vals = O.gather(fields, enc_z, w_z)
    good = good and vals.numpy().tobytes() == ref_vals.tobytes() and vals2.numpy().tobytes() == ref_vals.tobytes()
    print(f"case {case:3d} n={n:2d} N={npts:6d} k={k:2d} C={ncomp} lazy={int(lazy)} margin={margin} "
          f"nfailed={nf:6d} fallback={(status >= k).sum():5d} -> {'ok' if good else 'MISMATCH'}", flush=True)
    if not good:
        sys.exit(1)
print(f"{ncases} cases ok in {time.time() - t_start:.0f} s")
