"""kNN stage on strongly graded clouds (coordinates = uniform^p: density varies by orders of magnitude
across the domain) and on a cloud with a locally refined region -- what the density levels of the
search grid buy (MM_KNN_LEVELS=1 switches them off)."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd.device import Context

n = 4_000_000
rng = np.random.default_rng(0)
ctx = Context(0)
ctx.set_profiling(True)
out = {}
for power in (1.0, 1.5, 2.0, 3.0):
    src, tgt = rng.uniform(size=(n, 3)) ** power, rng.uniform(size=(n, 3)) ** power
    d_src, d_tgt = ctx.to_device(src), ctx.to_device(tgt)
    tree = ctx.knn_build(d_src)
    for _ in range(2):
        idx = tree.query(d_tgt, 20)
        t = ctx.last_timings()
    out[f"power_{power}"] = {"knn_query_ms": round(t["knn_query"], 2), "fast_kernel_ms": round(t["knn_cell"], 2)}
# a locally refined mesh: half of the points in a region refined 3x per axis (27x the density)
def two_density(m):
    return np.concatenate([rng.uniform(size=(m // 2, 3)), 0.35 + 0.2 * rng.uniform(size=(m - m // 2, 3))])
src, tgt = two_density(n), two_density(n)
d_src, d_tgt = ctx.to_device(src), ctx.to_device(tgt)
tree = ctx.knn_build(d_src)
for k in (8, 20):
    for _ in range(2):
        idx = tree.query(d_tgt, k)
        t = ctx.last_timings()
    out[f"refined_region_k{k}"] = {"knn_query_ms": round(t["knn_query"], 2), "fast_kernel_ms": round(t["knn_cell"], 2)}
print(json.dumps(out))
