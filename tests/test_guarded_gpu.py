"""The library under GUARDED allocations (MM_GUARD_ALLOC=1, multimesh_amd/csrc/mm_context.hip): every device
allocation -- the index, every scratch carve, the arrays the Python layer makes for NumPy inputs and for results
-- ends at the end of its mapping with unmapped addresses behind it, so a kernel that reads or writes past an
array faults at once instead of now and then.  Regression test of the fault found by the long fuzz runs of round
3 (the strip and cell kernels' staging loaded record `nsrc` of an index of nsrc records for the empty cells behind
the last source; it faulted only when nsrc * 32 bytes ended a mapping -- case 1365 of `fuzz_knn.py 1500 31003`:
17280 = 135 * 128 clustered sources), and a net under the other entry points.

The switch is read once per process, so the checks run in a child process (which also keeps a fault, should one
come back, out of the test runner's own GPU context)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_CHECKS = r"""
import sys
import numpy as np
sys.path.insert(0, ".")
from multimesh_amd import synth
from multimesh_amd.device import Context
from oracle import oracle as O

ctx = Context(0)
rng = np.random.default_rng(31003)

# 1. the geometry of the fault: clustered sources whose count is a multiple of 128 (nsrc * 32 bytes = whole pages),
#    targets spread over the box (many strips border empty cells behind the last source), short lists
for nsrc, ntgt, k in [(17280, 83868, 3), (4096, 20000, 8), (128 * 40, 7000, 20), (128, 500, 4)]:
    centres = rng.uniform(size=(7, 3))
    src = centres[rng.integers(0, len(centres), size=nsrc)] + rng.normal(scale=0.03, size=(nsrc, 3))
    lo, hi = src.min(axis=0), src.max(axis=0)
    tgt = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(ntgt, 3))
    idx = ctx.knn_build(src).query(tgt, k).numpy()
    assert np.array_equal(idx, O.knn_ckdtree(src, tgt, k, workers=-1)[0]), ("knn", nsrc, ntgt, k)

# 2. uniform clouds through the lane, strip and cell kernels' usual shapes (2-D and 1-D included)
for dim, nsrc, ntgt, k in [(3, 128 * 500, 150_000, 8), (3, 128 * 100, 3000, 20), (2, 128 * 64, 9000, 16), (1, 1280, 700, 5)]:
    src = rng.uniform(size=(nsrc, dim))
    tgt = rng.uniform(-0.1, 1.1, size=(ntgt, dim))
    idx = ctx.knn_build(src).query(tgt, k).numpy()
    ref = O.knn_ckdtree(src, tgt, k, workers=-1)[0].reshape(ntgt, k)
    assert np.array_equal(idx, ref), ("knn", dim, nsrc, ntgt, k)

# 3. the fused hex8 pass (values, operator, failures) against the oracle
pa, ca = synth.hex_mesh(17, seed=1, jitter=0.2)
pb, _ = synth.hex_mesh(21, seed=7, jitter=0.2)
pb = np.concatenate([pb, rng.uniform(-0.2, 1.2, size=(500, 3))])
fields = synth.vector_field(pa)[:2]
vals, nfailed = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20)
cen = O.centroid(ca, pa)
nn = O.knn_ckdtree(cen, pb, 20)[0]
enc, w, nf = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
assert nfailed == nf and np.array_equal(vals.numpy(), O.gather_numpy(fields, enc, w)), "fused hex8 pass"

# 4. unique points and the GLL path
g = synth.gll_mesh(5, 4, seed=3, jitter=0.15)
pts = g.reshape(-1, 3)
u, inv = ctx.unique_points(pts)
ur, ir = np.unique(pts, axis=0, return_inverse=True)
assert np.array_equal(u.numpy(), ur) and np.array_equal(inv.numpy().ravel(), ir.ravel()), "unique_points"
src_g = synth.gll_mesh(4, 4, seed=1, jitter=0.15)
f = rng.uniform(size=(1, src_g.shape[0], src_g.shape[1]))
out, missing = ctx.interpolate_gll(4, src_g, ur[::3], f, nelem_to_search=8)
assert out.numpy().shape[0] == len(ur[::3]) and missing >= 0
# 5. round 4's entry points: the fused pass in MM_FP_TOL (fast locate instance), a resident source, the order-free
#    unique, the legacy symbol (sizes found on the device)
ctx.set_fp_mode("tol")
vals_t, enc_t, w_t, nf_t = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
ok = w.any(axis=1)
assert nf_t == nf and np.array_equal(enc_t.numpy()[ok], enc[ok]) and np.abs(w_t.numpy()[ok] - w[ok]).max() <= 1e-12, "MM_FP_TOL"
ctx.set_fp_mode("exact")
src_h = ctx.source(pa, ca)
v_r, nf_r = src_h.interpolate(pb, fields, nelem_to_search=20)
assert nf_r == nf and np.array_equal(v_r.numpy(), vals.numpy()), "resident source"
src_h.free()
u2, inv2 = ctx.unique_points(pts, ordered=False)
assert len(u2.numpy()) == len(ur) and np.array_equal(u2.numpy()[inv2.numpy()], pts), "unique_points_any_order"
from multimesh_amd import helpers
lib = helpers.load_lib()
nn64 = np.ascontiguousarray(nn, dtype=np.int64)
enc_l, w_l = np.zeros((len(pb), 8), np.int64), np.zeros((len(pb), 8))
nf_l = lib.triLinearInterpolator(20, len(pb), nn64, np.ascontiguousarray(synth.reorder_hex8(ca)), enc_l, np.ascontiguousarray(pa), w_l,
                                 np.ascontiguousarray(pb))
assert nf_l == nf and np.array_equal(enc_l, enc) and np.array_equal(w_l, w), "legacy symbol"
print("ok")
"""


def test_entry_points_under_guarded_allocations():
    env = dict(os.environ, MM_GUARD_ALLOC="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CHECKS], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1000:], r.stderr[-3000:])
