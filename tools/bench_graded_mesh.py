"""The fused hex8 pipeline on meshes refined towards a corner (node coordinates u -> u^p per axis):
the metric's 10M -> 10M shape with element sizes spanning orders of magnitude -- what the density
levels of the search grid buy end to end (MM_KNN_LEVELS=1 switches them off)."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
ctx = Context(0)
ctx.set_profiling(True)
out = {}
pa0, ca = synth.hex_mesh(n, seed=1, jitter=0.1)
pb0, _ = synth.hex_mesh(n, seed=7, jitter=0.1)
powers = [float(a) for a in sys.argv[2:]] or [1.0, 1.5, 2.2]
for power in powers:
    pa, pb = pa0 ** power, pb0 ** power
    d_nodes, d_conn, d_pts = ctx.to_device(pa), ctx.to_device(ca), ctx.to_device(pb)
    d_f = ctx.to_device(synth.vector_field(pa)[:1])
    for _ in range(3):
        vals, nf = ctx.interpolate_hex8(d_nodes, d_conn, d_pts, d_f, nelem_to_search=20)
        t = ctx.last_timings()
    total = sum(t[s] for s in ("centroid", "knn_build", "knn_query", "locate", "gather"))
    out[f"power_{power}"] = {"step_ms": round(total, 2), "nfailed": nf,
                             **{s: round(t[s], 2) for s in ("knn_build", "knn_query", "locate")},
                             "max_err_linear": float(np.abs(vals.numpy()[:, 0] - synth.field_linear(pb)).max())}
print(json.dumps(out))
