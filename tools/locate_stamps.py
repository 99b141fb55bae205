"""Diagnostic (library built with `make EXTRA=-DMM_LOCATE_STAMPS`): per-phase share of a locate_pass_kernel wave's
cycles on the metric workload."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
pa, ca = synth.hex_mesh(n, seed=1); pb, _ = synth.hex_mesh(n, seed=7)
ctx = Context(0); ctx.set_profiling(True)
f = synth.vector_field(pa)[:1]
dn, dc, dp, df = (ctx.to_device(x) for x in (pa, ca, pb, f))
buf = (C.c_ulonglong * 16)()
for rep in range(3):
    ctx.interpolate_hex8(dn, dc, dp, df)
    t = ctx.last_timings()
    ctx.lib.mm_debug_locate_stamps(buf, 1)
    print("locate_pass0 ms (this diagnostic build):", t["locate_pass0"])
v = np.array(list(buf), dtype=np.float64)
names = ["round selection, queue reads", "target coordinates", "candidate + corners + box test", "Newton", "weights", "emit (field gathers, stores)", "queue appends"]
waves = v[15]; tot = v[:7].sum()
print(f"waves {int(waves)}, ticks per wave {tot / waves:.0f}")
for nm, x in zip(names, v[:7]):
    print(f"  {nm:32s} {x / waves:10.0f} ticks  {100 * x / tot:5.1f} %")
