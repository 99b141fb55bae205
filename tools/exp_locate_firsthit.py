"""How much of the locate stage is retries: the metric mesh with targets = the element centroids
(every target is accepted at its first candidate) against the metric targets (1.6 solves per target)."""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
pa, ca = synth.hex_mesh(216, seed=1)
pb, _ = synth.hex_mesh(216, seed=7)
cen = pa[ca].mean(axis=1)
ctx = Context(0); ctx.set_profiling(True)
d_nodes, d_conn, d_f = ctx.to_device(pa), ctx.to_device(ca), ctx.to_device(synth.vector_field(pa)[:1])
out = {}
for name, tgt in (("metric_targets", pb), ("centroid_targets", cen)):
    d_t = ctx.to_device(np.ascontiguousarray(tgt))
    for _ in range(3):
        vals, nf = ctx.interpolate_hex8(d_nodes, d_conn, d_t, d_f, nelem_to_search=20)
    out[name] = {"n": len(tgt), "nfailed": nf, **{k: round(v, 3) for k, v in ctx.last_timings().items() if v > 0}}
print(json.dumps(out, indent=1))
