"""Stand-alone A9 (mm_gather) at the metric size: 10 M targets, hex8 rows, C = 1 -- the HBM-bound
kernel that callers keeping the operator run per field (the fused values-only path forms the sum
inside the locate).  Prints one JSON line; used for the gather entry of the traffic profile."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd.device import Context

from multimesh_amd import synth

pa, ca = synth.hex_mesh(216, seed=1)            # the metric's source mesh
n, m = len(pa), len(pa)
rng = np.random.default_rng(0)
# operator rows as the locate stage produces them: target t sits in an element next to node t
elem = np.minimum(np.arange(n, dtype=np.int64) * len(ca) // n, len(ca) - 1)
ids = np.ascontiguousarray(synth.reorder_hex8(ca)[elem])
w = rng.uniform(size=(n, 8))
f = rng.normal(size=(1, m))
ctx = Context(0)
d_ids, d_w, d_f = ctx.asdevice(ids, np.int64), ctx.asdevice(w, np.float64), ctx.asdevice(f, np.float64)
ctx.set_profiling(True)
for _ in range(3):
    out = ctx.gather(d_f, d_ids, d_w)
ms = []
for _ in range(10):
    out = ctx.gather(d_f, d_ids, d_w)
    ms.append(ctx.last_timings()["gather"])
t = float(np.median(ms))
bytes_ = n * (128 + 72)
print(json.dumps({"kernel": "gather8_kernel<true>", "targets": n, "ms": t, "algorithmic_bytes": bytes_,
                  "achieved_GBps": bytes_ / t / 1e6, "frac_of_8TBps": bytes_ / t / 1e6 / 8000}))
