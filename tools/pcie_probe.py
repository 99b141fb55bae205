"""What the host link gives on this box: pageable vs pinned (registered) H2D / D2H rates and the cost of
page-locking the caller's arrays -- the numbers behind DESIGN.md's host-array boundary section."""
import time, sys
import numpy as np
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 636_000_000
a = np.ones(n // 8, dtype=np.int64)
d = torch.empty(n // 8, dtype=torch.int64, device="cuda")
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
ta = torch.from_numpy(a)
x = t(lambda: d.copy_(ta)); print(f"pageable H2D {n/1e6:.0f} MB: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
h = torch.empty(n // 8, dtype=torch.int64)
x = t(lambda: h.copy_(d)); print(f"pageable D2H: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); treg = time.perf_counter() - t0
print(f"hipHostRegister rc={rc}: {treg*1e3:.1f} ms = {n/treg/1e9:.1f} GB/s")
x = t(lambda: d.copy_(ta, non_blocking=True)); print(f"registered H2D: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
t0 = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); print(f"unregister {1e3*(time.perf_counter()-t0):.1f} ms")
p = torch.empty(n // 8, dtype=torch.int64).pin_memory()
x = t(lambda: d.copy_(p, non_blocking=True)); print(f"pinned H2D: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
x = t(lambda: p.copy_(d, non_blocking=True)); print(f"pinned D2H: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
x = t(lambda: p.copy_(ta)); print(f"host memcpy pageable->pinned (1 thread): {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
torch.set_num_threads(16)
x = t(lambda: p.copy_(ta)); print(f"host memcpy pageable->pinned (torch, 16 threads): {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
# two streams, halves
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
half = n // 16
def two():
    with torch.cuda.stream(s1): d[:half].copy_(p[:half], non_blocking=True)
    with torch.cuda.stream(s2): d[half:].copy_(p[half:], non_blocking=True)
x = t(two); print(f"pinned H2D on two streams: {x*1e3:.1f} ms = {n/x/1e9:.1f} GB/s")
