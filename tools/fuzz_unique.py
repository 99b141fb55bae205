"""Randomised check of mm_unique_points against np.unique(axis=0, return_inverse=True), and of the order-free form
mm_unique_points_any_order: the same SET of unique rows, in the order of their first occurrence, and an inverse that
rebuilds the input."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd.device import Context

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ctx = Context(0)
t0 = time.time()
for case in range(ncases):
    dim = int(rng.integers(1, 4))
    n = int(rng.choice([rng.integers(1, 50), rng.integers(50, 5000), rng.integers(5000, 1_500_000)]))
    kind = rng.choice(["floats", "few_values", "lattice", "signed", "last_bits"])
    if kind == "floats":
        base = rng.normal(size=(max(1, n // int(rng.integers(1, 6))), dim))
        pts = base[rng.integers(0, len(base), size=n)]
    elif kind == "few_values":
        pts = rng.integers(-3, 4, size=(n, dim)).astype(np.float64)
    elif kind == "lattice":
        pts = np.round(rng.uniform(-5, 5, size=(n, dim)), int(rng.integers(0, 3)))
    elif kind == "last_bits":
        # few values of x, each spread over its last bits (the main sort orders by the top 48 bits of x only)
        pts = rng.uniform(-2, 2, size=(n, dim))
        vals = rng.uniform(-2, 2, size=int(rng.integers(1, 40)))
        pts[:, 0] = vals[rng.integers(0, len(vals), size=n)] * (1.0 + rng.integers(0, int(rng.choice([2, 50, 70000])), size=n) * 2.0 ** -52)
        pts = pts[rng.integers(0, n, size=n)]
    else:
        pts = rng.integers(-2, 3, size=(n, dim)).astype(np.float64) * rng.choice([1.0, -0.0, 1e-300, 1e300], size=(n, dim))
    pts = np.ascontiguousarray(pts)
    u, inv = ctx.unique_points(pts)
    u, inv = u.numpy(), inv.numpy()
    ru, rinv = np.unique(pts, axis=0, return_inverse=True)
    good = u.shape == ru.shape and np.array_equal(u, ru) and np.array_equal(inv, rinv.reshape(-1)) and np.array_equal(u[inv], pts)
    # the order-free form: first occurrences, ascending
    u2, inv2 = ctx.unique_points(pts, ordered=False)
    u2, inv2 = u2.numpy(), inv2.numpy()
    first = np.full(len(ru), n, dtype=np.int64)
    np.minimum.at(first, rinv.reshape(-1), np.arange(n))
    order = np.argsort(first)                       # np.unique's classes by first occurrence
    want_u2 = pts[first[order]] + 0.0               # (-0.0 is stored as +0.0)
    pos = np.empty(len(ru), dtype=np.int64)
    pos[order] = np.arange(len(ru))
    good2 = (u2.shape == ru.shape and np.array_equal(u2, want_u2) and not np.signbit(u2[u2 == 0]).any()
             and np.array_equal(inv2, pos[rinv.reshape(-1)]) and np.array_equal(u2[inv2], pts))
    good = good and good2
    print(f"case {case:3d} dim={dim} n={n:7d} {kind:10s} unique={len(ru):7d} -> {'ok' if good else 'MISMATCH'}", flush=True)
    if not good:
        sys.exit(1)
print(f"{ncases} cases ok in {time.time() - t0:.0f} s")
