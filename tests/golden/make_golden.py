#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/*.npz.

Run in a container that has /root/reference:   python tests/golden/make_golden.py

Every expected output below comes from the REFERENCE, not from this repository's code:

* centroid / enc / w / nfailed: the reference's own C files (src/centroid.c,
  src/trilinearinterpolator.c) compiled into oracle/_ref/multi_mesh_ref.so by
  oracle/Makefile and driven exactly as reference scripts/cli.py:62-100 does
  (zero-initialised outputs, exodus->locator column reorder, k nearest centroids).
* nn (and its distances): scipy.spatial.cKDTree(cen, balanced_tree=False).query(pts, k),
  the third-party call the reference makes at scripts/cli.py:66-73 (scipy version recorded
  in each file).
* values: the literal NumPy statement np.sum(field[enc] * w, axis=1) (scripts/cli.py:100).

Fixtures hold arrays only (inputs + expected outputs); no reference source text.
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from multimesh_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

META = dict(scipy_version=scipy.__version__, numpy_version=np.__version__,
            generator="tests/golden/make_golden.py")


def run_reference(points_a, conn_exodus, points_b, k, fields):
    """The reference pipeline scripts/cli.py:62-100 on arrays."""
    cen = O.ref_centroid(conn_exodus, points_a)
    nn, dist = O.knn_ckdtree(cen, points_b, k)
    # the reference's own statement (scripts/cli.py:79-81), written out here so that the A3 fixture does
    # not depend on this repository's synth.reorder_hex8
    permutation = [0, 3, 2, 1, 4, 5, 6, 7]
    i = np.argsort(permutation)
    conn = np.ascontiguousarray(conn_exodus[:, i])
    enc, w, nfailed = O.ref_locate_hex8(nn, conn, points_a, points_b)
    vals = O.gather_numpy(fields, enc, w)
    return dict(centroid=cen, nn=nn, nn_dist=dist, conn_reordered=conn, enc=enc, w=w,
                nfailed=np.int64(nfailed), values=vals)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays, **{"meta_" + k: np.array(v) for k, v in META.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} kB")


def case_small():
    # general position, every target inside the source hull -> nfailed == 0
    pa, ca = synth.hex_mesh(8, seed=1)
    pb, _ = synth.hex_mesh(9, seed=7)
    fields = np.stack([synth.field_linear(pa), synth.field_smooth(pa)])
    out = run_reference(pa, ca, pb, 20, fields)
    assert out["nfailed"] == 0
    save("hex8_small.npz", points_a=pa, conn_a=ca, points_b=pb, k=np.int64(20), fields=fields,
         **out)


def case_hard():
    # strongly distorted source (jitter 0.42 h), targets blown up 8 % around the centre so that
    # the shell outside the hull exercises: accept < 1.025, the "< 1.5 after the last candidate"
    # fallback, and outright failures; small k makes the last-candidate branch frequent.
    pa, ca = synth.hex_mesh(7, seed=3, jitter=0.42)
    rng = np.random.default_rng(11)
    pb = rng.uniform(-0.08, 1.08, size=(700, 3))
    # a few far-away points (hull check |xi| <= 2 fails or Newton does not converge)
    pb[:20] = rng.uniform(-1.0, 2.0, size=(20, 3))
    # and the source nodes themselves (xi exactly on corners / faces)
    pb = np.concatenate([pb, pa[::5]])
    fields = np.stack([synth.field_linear(pa), synth.field_smooth(pa)])
    for k in (1, 3, 20):
        out = run_reference(pa, ca, pb, k, fields)
        print(f"  hard k={k}: nfailed={int(out['nfailed'])}")
        save(f"hex8_hard_k{k}.npz", points_a=pa, conn_a=ca, points_b=pb, k=np.int64(k),
             fields=fields, **out)


def case_structured_ties():
    # unjittered grid, targets = the source nodes: kNN is tie-laden (cKDTree's tie order is
    # traversal dependent), so locate parity uses the stored nn as its input.
    pa, ca = synth.hex_mesh(6, seed=1, jitter=0.0)
    pb = pa.copy()
    fields = np.stack([synth.field_linear(pa)])
    out = run_reference(pa, ca, pb, 20, fields)
    save("hex8_structured.npz", points_a=pa, conn_a=ca, points_b=pb, k=np.int64(20),
         fields=fields, **out)


def case_knn():
    rng = np.random.default_rng(5)
    arrays = {}
    # 3-D general position, queries partly outside the source bounding box
    src3 = rng.uniform(0, 1, size=(3000, 3))
    q3 = rng.uniform(-0.2, 1.2, size=(400, 3))
    for k in (1, 5, 20):
        idx, d = O.knn_ckdtree(src3, q3, k)
        arrays[f"idx3_k{k}"] = idx
        arrays[f"dist3_k{k}"] = d
    # anisotropic cloud (graded density) -> exercises ring expansion in a uniform grid
    srcg = np.concatenate([rng.normal(0.5, 0.02, size=(1500, 3)), rng.uniform(0, 1, size=(500, 3))])
    qg = rng.uniform(0, 1, size=(300, 3))
    idx, d = O.knn_ckdtree(srcg, qg, 20)
    arrays.update(srcg=srcg, qg=qg, idxg_k20=idx, distg_k20=d)
    # 2-D (cfg1 is a 2-D mesh)
    src2 = rng.uniform(0, 1, size=(2000, 2))
    q2 = rng.uniform(-0.1, 1.1, size=(300, 2))
    idx, d = O.knn_ckdtree(src2, q2, 20)
    arrays.update(src2=src2, q2=q2, idx2_k20=idx, dist2_k20=d)
    # fewer sources than k: cKDTree pads with index == nsrc and distance inf
    srcs = rng.uniform(0, 1, size=(7, 3))
    qs = rng.uniform(0, 1, size=(16, 3))
    idx, d = O.knn_ckdtree(srcs, qs, 20)
    arrays.update(srcs=srcs, qs=qs, idxs_k20=idx, dists_k20=d)
    save("knn.npz", src3=src3, q3=q3, **arrays)


def case_gather():
    # NumPy's row-sum order for the element sizes the reference uses (P = 8, 25, 27, 125)
    rng = np.random.default_rng(9)
    arrays = {}
    for P in (4, 8, 25, 27, 125):
        field = rng.normal(size=(2, 500)) * 10.0 ** rng.integers(-3, 4, size=(2, 500))
        ids = rng.integers(0, 500, size=(64, P)).astype(np.int64)
        w = rng.normal(size=(64, P))
        arrays[f"field_P{P}"] = field
        arrays[f"ids_P{P}"] = ids
        arrays[f"w_P{P}"] = w
        arrays[f"values_P{P}"] = O.gather_numpy(field, ids, w)
    save("gather.npz", **arrays)


if __name__ == "__main__":
    if not O.have_reference():
        O.build(ref=True)
    assert O.have_reference(), "needs oracle/_ref/multi_mesh_ref.so (i.e. /root/reference mounted)"
    case_small()
    case_hard()
    case_structured_ties()
    case_knn()
    case_gather()
