"""Experiment: how much idle capacity is there for concurrent kernels?  Two contexts (two HIP streams), each
running the whole pipeline on its own resident copy of the metric workload, from two host threads, against the
same number of steps run one after the other."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import torch
from multimesh_amd import synth
from multimesh_amd.device import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
steps = 20
pa, ca = synth.hex_mesh(n, seed=1)
f = synth.vector_field(pa)[:1]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs, data = [], []
for s, seed in zip(streams, (7, 8)):
    c = Context(0, stream=s.cuda_stream)
    pb, _ = synth.hex_mesh(n, seed=seed)
    d = [c.to_device(x) for x in (pa, ca, pb, f)]
    out = c.empty((len(pb), 1), np.float64)
    ctxs.append(c); data.append((d, out))
def run(i, k):
    c, (d, out) = ctxs[i], data[i]
    for _ in range(k):
        c.interpolate_hex8(*d, out=out)
for i in range(2): run(i, 3)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(0, steps); run(1, steps); torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, steps)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); t_par = time.perf_counter() - t0
print(f"{2*steps} steps one after the other: {t_seq*1e3/(2*steps):.3f} ms/step; two streams at once: {t_par*1e3/(2*steps):.3f} ms/step")
