"""(GPU box) the guessed-grid miss of tests/test_grid_guess_gpu.py::test_a_miss_at_a_realistic_size_is_cheap alone, for a
kernel trace:  rocprofv3 --kernel-trace --stats -d out -o p -- python3 tools/dbg_guess_miss.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from multimesh_amd import synth
from multimesh_amd.device import Context
pa, ca = synth.hex_mesh(101, seed=1)
pb, _ = synth.hex_mesh(101, seed=7)
f = synth.vector_field(pa)[:1]
pa2, pb2 = pa * 3.0 + 10.0, pb * 3.0 + 10.0
f2 = synth.vector_field(pa2)[:1]
c = Context(0)
d = [c.to_device(x) for x in (pa, ca, pb, f, pa2, pb2, f2)]
for _ in range(3):
    c.interpolate_hex8(d[0], d[1], d[2], d[3])
c.synchronize()
t0 = time.perf_counter()
c.interpolate_hex8(d[4], d[1], d[5], d[6])
c.synchronize()
print("miss call", time.perf_counter() - t0)
