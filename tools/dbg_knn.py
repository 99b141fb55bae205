"""Diagnostic: compare the kNN fast path against cKDTree and describe the mismatching rows."""
import sys
import numpy as np
from scipy.spatial import cKDTree
from multimesh_amd.device import Context

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(k)
src = rng.uniform(size=(200_000, 3))
q = rng.uniform(-0.05, 1.05, size=(50_000, 3))
ctx = Context(0)
idx = ctx.knn_build(src).query(q, k).numpy()
d, ref = cKDTree(src).query(q, k=k, workers=-1)
bad = np.nonzero((idx != ref).any(axis=1))[0]
print("mismatching rows", len(bad), "of", len(q))
for r in bad[:8]:
    same_set = set(idx[r]) == set(ref[r])
    dd = np.sqrt(((src[np.clip(idx[r], 0, len(src) - 1)] - q[r]) ** 2).sum(1))
    print(r, "same set" if same_set else "different set", "sorted" if (np.diff(dd) >= 0).all() else "unsorted")
    print("  ours", idx[r][:12], "\n  ref ", ref[r][:12])
    print("  d ours", np.round(dd[:8], 5), "ref", np.round(d[r][:8], 5), "q", q[r])
