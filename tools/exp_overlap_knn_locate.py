"""Experiment: do the kNN lane kernel and the locate pass kernel fill each other's bubbles when they run at the
same time (two HIP streams)?  Staged calls on the metric workload: N kNN queries (k = 8) and N locate calls, one
after the other vs concurrently from two host threads."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import torch
from multimesh_amd import synth
from multimesh_amd.device import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
steps = 15
pa, ca = synth.hex_mesh(n, seed=1); pb, _ = synth.hex_mesh(n, seed=7)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c1, c2 = Context(0, stream=s1.cuda_stream), Context(0, stream=s2.cuda_stream)
cen = c1.centroid(ca, pa)
tree = c1.knn_build(cen)
p1 = c1.to_device(pb)
nn = tree.query(p1, 8)
nn2 = c2.to_device(nn.numpy())
d2 = [c2.to_device(x) for x in (ca, pa, pb)]
enc, w = c2.zeros((len(pb), 8), np.int64), c2.zeros((len(pb), 8), np.float64)
idx = c1.empty((len(pb), 8), np.int64)
from multimesh_amd.helpers import check
def knn(k):
    for _ in range(k): check(c1.lib.mm_knn_query(c1.handle, tree.handle, p1.ptr, len(pb), 8, idx.ptr, None), "q")
    c1.synchronize()
def loc(k):
    for _ in range(k): c2.locate_hex8(nn2, d2[0], d2[1], d2[2], enc=enc, weights=w, conn_is_exodus=True)
knn(2); loc(2); torch.cuda.synchronize()
t0 = time.perf_counter(); knn(steps); torch.cuda.synchronize(); tk = time.perf_counter() - t0
t0 = time.perf_counter(); loc(steps); torch.cuda.synchronize(); tl = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=knn, args=(steps,)), threading.Thread(target=loc, args=(steps,))]
[t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); tb = time.perf_counter() - t0
print(f"kNN query alone {tk*1e3/steps:.3f} ms, locate alone {tl*1e3/steps:.3f} ms, sum {1e3*(tk+tl)/steps:.3f}; both at once {tb*1e3/steps:.3f} ms per pair")
