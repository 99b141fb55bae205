#!/usr/bin/env python3
"""MM_FP_TOL soundness experiment on the host: newton_hex8_fast (tests/host build of csrc/mm_newton_hex8.h) against the
oracle's reference iteration over random solves of every kind; prints certified / unsure shares, wrong verdicts (must
be 0) and the largest |xi_fast - xi_ref| in units of the margin delta.  usage: exp_fast_newton.py [n per case] [cap]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from tests.test_newton_host import CASES, OUT, elements  # noqa: E402


def metric_like(rng, n, nside=216):
    """The solves the metric workload's locate pass runs: a slab of the 216^3 lattice (same spacing and coordinates),
    source seed 1 / target seed 7 jitter, candidates = the 8 nearest centroids in cKDTree order, walked until the
    reference accepts one (the x/y box skip of the pass kernel is not applied: a few more rejected solves)."""
    from scipy.spatial import cKDTree
    from multimesh_amd import synth
    m = 40
    h = 1.0 / (nside - 1)
    src, conn = synth.hex_mesh(m, seed=1)
    tgt, _ = synth.hex_mesh(m, seed=7)
    scale = (m - 1) * h
    off = np.array([0.61, 0.33, 0.47])
    src = src * scale + off
    tgt = tgt * scale + off
    connr = synth.reorder_hex8(conn)
    cen = O.centroid(connr, src)
    _, nn = cKDTree(cen, balanced_tree=False).query(tgt, k=8)
    L = O.lib()
    pn, vt = [], []
    xi = np.zeros(3)
    for i in rng.permutation(len(tgt))[: max(1, n // 2)]:
        for j in range(8):
            v = np.ascontiguousarray(src[connr[nn[i, j]]])
            pn.append(tgt[i])
            vt.append(v)
            if L.mmo_hex8_newton(np.ascontiguousarray(tgt[i]), v, xi, None) and np.abs(xi).max() < 1.025:
                break
    return np.ascontiguousarray(np.array(pn)), np.ascontiguousarray(np.array(vt))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    L = C.CDLL(OUT)
    f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags=["C_CONTIGUOUS"])
    L.nh_fast_stats.restype = None
    L.nh_fast_stats.argtypes = [C.c_int64, f64p, f64p, C.c_void_p, C.c_int, np.ctypeslib.ndpointer(dtype=np.int64), f64p]
    fn = C.cast(O.lib().mmo_hex8_newton, C.c_void_p)
    rng = np.random.default_rng(4242)
    sets = [("metric-like", metric_like(rng, n))]
    for c, (jit, scale, off, spread) in enumerate(CASES):
        sets.append((f"case{c} jit={jit} scale={scale:g} spread={spread}", elements(rng, n, jit, scale, off, spread)))
    for name, (pnt, vtx) in sets:
        out = np.zeros(6, np.int64)
        dout = np.zeros(4)
        L.nh_fast_stats(len(pnt), pnt, vtx, fn, cap, out, dout)
        tot = out[:3].sum()
        print(f"{name:45s} accept {out[0]/tot:6.3f} reject {out[1]/tot:6.3f} unsure {out[2]/tot:7.4f} (of them ref-accepts {out[5]})"
              f" WRONG {out[3]} tripdiff {out[4]} | max d/delta {dout[0]:.3g} max d {dout[1]:.3g} max delta {dout[2]:.3g} max ratio {dout[3]:.3g}")


if __name__ == "__main__":
    main()
