#!/usr/bin/env python3
"""cfg5-shaped timing of the GLL path (order-4 hexes): 43^3 source elements, targets = the unique GLL
points of a 47^3-element mesh.  Not the headline bench; prints stage times."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimesh_amd import synth
from multimesh_amd.device import Context

ap = argparse.ArgumentParser()
ap.add_argument("--n-src", type=int, default=44)
ap.add_argument("--n-tgt", type=int, default=48)
ap.add_argument("--order", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
src = synth.gll_mesh(a.n_src, a.order, seed=1)
tgt_en = synth.gll_mesh(a.n_tgt, a.order, seed=7).reshape(-1, 3)
t0 = time.time(); tgt = np.unique(tgt_en, axis=0); t_unique = time.time() - t0
fields = synth.field_smooth(src.reshape(-1, 3)).reshape(1, *src.shape[:2])
ctx = Context(0); ctx.set_profiling(True)
d_en = ctx.to_device(tgt_en)
for rep in range(a.reps):   # A11 on the device: the same unique set and inverse index as np.unique
    t0 = time.perf_counter(); d_u, d_inv = ctx.unique_points(d_en); ctx.synchronize(); t_dev_unique = time.perf_counter() - t0
assert np.array_equal(d_u.numpy(), tgt)
d_src, d_tgt, d_f = ctx.to_device(src), ctx.to_device(tgt), ctx.to_device(fields)
cen = ctx.to_device(src.mean(axis=1))
res = {}
for rep in range(a.reps):
    t0 = time.perf_counter(); tree = ctx.knn_build(cen); ctx.synchronize(); res["knn_build_ms"] = (time.perf_counter()-t0)*1e3
    t0 = time.perf_counter(); nn = tree.query(d_tgt, 20); ctx.synchronize(); res["knn_query_ms"] = (time.perf_counter()-t0)*1e3
    elem = co = None   # release the previous 1 kB-per-target coefficient array before the next one is allocated
    t0 = time.perf_counter(); elem, co, miss = ctx.locate_gll(a.order, nn, d_src, d_tgt, 1.05, False); res["locate_wall_ms"] = (time.perf_counter()-t0)*1e3
    res["locate_ms"] = ctx.last_timings()["locate"]   # the stage itself; the wall time includes allocating the coefficient array
    t0 = time.perf_counter(); vals = ctx.gather_elem(d_f, elem, co); ctx.synchronize(); res["gather_ms"] = (time.perf_counter()-t0)*1e3
# the fused entry (values only: weighted sum formed where a target is accepted; lazy candidate lists)
for lazy in (False, True):
    ctx.set_lazy_lists(lazy)
    for rep in range(a.reps):
        t0 = time.perf_counter(); fv, fmiss = ctx.interpolate_gll(a.order, d_src, d_tgt, d_f, nelem_to_search=20); dt = (time.perf_counter()-t0)*1e3
    key = "fused_lazy" if lazy else "fused_eager"
    res[key + "_ms"] = round(dt, 3)
    res[key + "_stages"] = {k: round(v, 3) for k, v in ctx.last_timings().items() if v > 0}
    assert fmiss == miss and np.array_equal(fv.numpy(), vals.numpy())
res["fused_points_per_s"] = len(tgt) / (res["fused_lazy_ms"] * 1e-3)
v = vals.numpy()[:, 0]
res.update(n_src_elem=int(src.shape[0]), P=int(src.shape[1]), n_targets=int(len(tgt)), missing=int(miss),
           host_unique_s=round(t_unique, 2), device_unique_ms=round(t_dev_unique * 1e3, 2), element_nodal_points=int(len(tgt_en)), max_err=float(np.abs(v - synth.field_smooth(tgt)).max()))
res["points_per_s"] = len(tgt) / ((res["knn_build_ms"] + res["knn_query_ms"] + res["locate_ms"] + res["gather_ms"]) * 1e-3)
print(json.dumps(res))
