"""Randomised check of mm_knn_build / mm_knn_query against scipy's cKDTree: 1-3 dimensions, uniform,
clustered, anisotropic and lattice-like clouds, targets inside and outside the sources' box, k from 1
to 64, more targets than sources and the reverse.  General position (rows that differ only in the order
of exactly equidistant sources are accepted: ours is by index, cKDTree's unspecified).  Not part of the
test suite; exits non-zero on the first mismatch."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from scipy.spatial import cKDTree
from multimesh_amd.device import Context

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
start = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # skip the cases before this one (same random sequence)
ctx = Context(0)
t0 = time.time()
for case in range(ncases):
    dim = int(rng.choice([1, 2, 3, 3, 3]))
    nsrc = int(rng.choice([rng.integers(1, 200), rng.integers(200, 20_000), rng.integers(20_000, 400_000)]))
    ntgt = int(rng.choice([rng.integers(1, 300), rng.integers(300, 120_000)]))
    kind = rng.choice(["uniform", "clustered", "aniso", "lattice", "graded"])
    if kind == "uniform":
        src = rng.uniform(size=(nsrc, dim))
    elif kind == "clustered":
        centres = rng.uniform(size=(int(rng.integers(1, 12)), dim))
        src = centres[rng.integers(0, len(centres), size=nsrc)] + rng.normal(scale=rng.uniform(1e-3, 0.1), size=(nsrc, dim))
    elif kind == "aniso":
        src = rng.uniform(size=(nsrc, dim)) * rng.uniform(1e-2, 50.0, size=dim) + rng.uniform(-1e4, 1e4, size=dim)
    elif kind == "lattice":
        m = max(2, int(round(nsrc ** (1.0 / dim))))
        g = np.stack(np.meshgrid(*[np.arange(m, dtype=np.float64)] * dim, indexing="ij"), axis=-1).reshape(-1, dim)
        src = g + rng.uniform(-0.2, 0.2, size=g.shape)
        nsrc = len(src)
    else:
        src = rng.uniform(size=(nsrc, dim)) ** rng.uniform(1.0, 2.5)
    lo, hi = src.min(axis=0), src.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    margin = rng.choice([0.0, 0.1, 1.0])
    tgt = rng.uniform(lo - margin * span, hi + margin * span, size=(ntgt, dim))
    if rng.random() < 0.3:                      # some targets are sources
        take = rng.integers(0, nsrc, size=min(ntgt, 200))
        tgt[: len(take)] = src[take]
    k = int(rng.choice([1, 2, 3, 4, 7, 8, 13, 16, 20, 24, 25, 31, 32, 33, 40, 64]))
    k = min(k, nsrc)
    want_dist = bool(rng.random() < 0.5)
    if (only >= 0 and case != only) or case < start:
        continue
    tree = ctx.knn_build(src)
    res = tree.query(tgt, k, want_dist=True) if want_dist else (tree.query(tgt, k), None)
    idx = res[0].numpy().reshape(ntgt, k)
    d_ref, i_ref = cKDTree(src).query(tgt, k=k, workers=-1)
    i_ref = i_ref.reshape(ntgt, k)
    good = np.array_equal(idx, i_ref)
    if not good:
        # exact ties (two sources at bit-equal distance -- it happens in 1-D): cKDTree's order among them is
        # unspecified, ours is by index; accept rows that differ only inside groups of equal distance
        d_ref2 = d_ref.reshape(ntgt, k)
        rows = np.nonzero((idx != i_ref).any(axis=1))[0]
        def tie_ok(r):
            d_ours = np.sqrt(((src[np.clip(idx[r], 0, nsrc - 1)] - tgt[r]) ** 2).sum(1))
            return np.array_equal(d_ours, d_ref2[r]) and sorted(idx[r]) == sorted(i_ref[r]) and \
                all(idx[r][j] == i_ref[r][j] or (j > 0 and d_ours[j] == d_ours[j - 1]) or (j + 1 < k and d_ours[j] == d_ours[j + 1])
                    for j in range(k))
        good = all(tie_ok(r) for r in rows)
        if good:
            print(f"  ({len(rows)} row(s) differ only in the order of equidistant sources)")
    if good and res[1] is not None:
        good = np.allclose(res[1].numpy().reshape(ntgt, k), d_ref.reshape(ntgt, k), rtol=4e-16, atol=0)
    print(f"case {case:4d} dim={dim} {kind:9s} nsrc={nsrc:6d} ntgt={ntgt:6d} k={k:2d} margin={margin} -> {'ok' if good else 'MISMATCH'}",
          flush=True)
    if not good:
        bad = np.nonzero((idx != i_ref).any(axis=1))[0]
        print("  rows differing:", len(bad), "first:", bad[:5])
        for r in bad[:3]:
            pos = np.nonzero(idx[r] != i_ref[r])[0]
            dd = np.sqrt(((src[np.clip(idx[r], 0, nsrc - 1)] - tgt[r]) ** 2).sum(1))
            print("   row", r, "at", pos[:6], "ours", idx[r][pos][:6], "ref", i_ref[r][pos][:6], "same set", set(idx[r]) == set(i_ref[r]),
                  "d ours", dd[pos][:4], "d ref", d_ref.reshape(ntgt, k)[r][pos][:4])
        sys.exit(1)
print(f"{ncases} cases ok in {time.time() - t0:.0f} s")
