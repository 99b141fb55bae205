"""MM_FP_TOL (mm_set_fp_mode): the hex8 locate stage in the cheaper Newton arithmetic of csrc/mm_newton_hex8.h.

The contract (include/multimesh_hip.h): node ids, the failed count and the rows of failed points are BIT-IDENTICAL to the
reference (every decision of the reference's iteration is certified with margin or the solve is repeated in the reference's
arithmetic); weights and interpolated values agree to TOL = max(1e-12, 64 eps max|x| / shortest element edge) -- 1e-12
on all meshes of this file -- weights absolutely (they are O(1)), values relative to 8 max|field|.  MM_FP_EXACT (the
default) stays at 0 ulp: that is every other test file."""
import os
import subprocess
import sys

import numpy as np
import pytest

from multimesh_amd import helpers, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from multimesh_amd.device import Context

    c = Context(0)
    assert c.fp_mode() == "exact"          # the default
    c.set_fp_mode("tol")
    assert c.fp_mode() == "tol"
    yield c
    c.close()


def close(a, b, scale=1.0):
    return np.abs(np.asarray(a) - np.asarray(b)).max(initial=0.0) <= TOL * scale


@pytest.mark.parametrize("name", ["hex8_small", "hex8_hard_k1", "hex8_hard_k3", "hex8_hard_k20"])
def test_golden_fixtures_ids_exact_values_within_tolerance(ctx, golden, name):
    d = golden(name)
    k = int(d["k"])
    fmax = 8 * np.abs(d["fields"]).max()
    failed = ~d["w"].any(axis=1)
    for lazy in (True, False):
        ctx.set_lazy_lists(lazy)
        try:
            vals, enc, w, nf = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"],
                                                    nelem_to_search=k, want_operator=True)
            vals2, nf2 = ctx.interpolate_hex8(d["points_a"], d["conn_a"], d["points_b"], d["fields"], nelem_to_search=k)
        finally:
            ctx.set_lazy_lists(True)
        assert nf == nf2 == int(d["nfailed"])
        assert np.array_equal(enc.numpy(), d["enc"])                       # bit-exact, always
        assert close(w.numpy(), d["w"]) and close(vals.numpy(), d["values"], fmax) and close(vals2.numpy(), d["values"], fmax)
        # rows of failed points come from the reference-order kernel: exact zeros, +0.0 values
        assert not w.numpy()[failed].any()
        assert vals2.numpy()[failed].tobytes() == np.ascontiguousarray(d["values"][failed]).tobytes()
    # the staged call
    enc, w, nf = ctx.locate_hex8(d["nn"], d["conn_reordered"], d["points_a"], d["points_b"])
    assert nf == int(d["nfailed"]) and np.array_equal(enc.numpy(), d["enc"]) and close(w.numpy(), d["w"])
    enc, w, nf = ctx.locate_hex8(d["nn"], d["conn_a"], d["points_a"], d["points_b"], conn_is_exodus=True)
    assert nf == int(d["nfailed"]) and np.array_equal(enc.numpy(), d["enc"]) and close(w.numpy(), d["w"])


def test_in_place_contract_holds(ctx, golden):
    d = golden("hex8_hard_k3")
    n = d["nn"].shape[0]
    enc0 = np.full((n, 8), 7, np.int64)
    w0 = np.full((n, 8), 0.25)
    enc, w, nf = ctx.locate_hex8(d["nn"], d["conn_reordered"], d["points_a"], d["points_b"],
                                 enc=ctx.to_device(enc0), weights=ctx.to_device(w0))
    failed = ~d["w"].any(axis=1)
    assert nf == failed.sum() == int(d["nfailed"])
    assert np.array_equal(enc.numpy()[failed], enc0[failed]) and np.array_equal(w.numpy()[failed], w0[failed])
    assert np.array_equal(enc.numpy()[~failed], d["enc"][~failed])


def test_targets_in_on_and_outside_a_distorted_mesh_vs_oracle(ctx):
    pa, ca = synth.hex_mesh(33, seed=21, jitter=0.3)
    rng = np.random.default_rng(22)
    pb = rng.uniform(-0.03, 1.03, size=(60_000, 3))
    pb[:500] = pa[rng.integers(0, len(pa), 500)]              # exactly on nodes: max|xi| = 1 in up to 8 elements
    conn = synth.reorder_hex8(ca)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc, w, nf = ctx.locate_hex8(nn, conn, pa, pb)
    enc_o, w_o, nf_o = O.locate_hex8(nn, conn, pa, pb)
    assert nf == nf_o and nf > 0
    assert np.array_equal(enc.numpy(), enc_o) and close(w.numpy(), w_o)
    stats = ctx.last_locate_stats()
    assert 0 < stats["redone_exact"] < 0.2 * len(pb)          # some solves are repeated exactly, most are not


def test_lists_that_run_out_sheared_mesh(ctx):
    pa, ca = synth.hex_mesh(24, seed=5, jitter=0.3)
    pa = pa.copy()
    pa[:, 0] += 0.9 * pa[:, 2] + 0.5 * pa[:, 1]
    pa[:, 2] *= 0.15
    rng = np.random.default_rng(11)
    pb = rng.uniform(pa.min(axis=0) - 0.02, pa.max(axis=0) + 0.02, size=(40_000, 3))
    fields = synth.vector_field(pa)[:2]
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, nelem_to_search=20, want_operator=True)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o, status = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb, want_status=True)
    ok = status >= 0
    assert nf == nf_o and nf_o > 0 and (status >= 8).sum() > 100
    # (flat sheared elements: max|x| / shortest edge ~ 1.4 / 0.0045 -> the stated tolerance is 64 eps * that = 4.4e-12)
    assert np.array_equal(enc.numpy()[ok], enc_o[ok]) and np.abs(w.numpy()[ok] - w_o[ok]).max() <= 5e-12
    assert not enc.numpy()[~ok].any() and not w.numpy()[~ok].any()


def test_earth_scale_coordinates(ctx):
    # coordinates in metres around 6e6 with 3e4 m elements: |x| / h = 200, like the BASELINE meshes
    pa, ca = synth.hex_mesh(30, seed=3, jitter=0.25)
    pa = pa * 9.0e5 + np.array([3.1e6, -2.2e6, 5.0e6])
    pb = np.random.default_rng(4).uniform(pa.min(axis=0), pa.max(axis=0), size=(50_000, 3))
    fields = np.ascontiguousarray(synth.vector_field((pa - pa.min(axis=0)) / 9.0e5)[:2])
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, want_operator=True)
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb, 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb)
    assert nf == nf_o == 0 and np.array_equal(enc.numpy(), enc_o)
    assert close(w.numpy(), w_o) and close(vals.numpy(), O.gather(fields, enc_o, w_o), 8 * np.abs(fields).max())


def test_cfg2_full_size_tol_against_exact_on_every_target(ctx):
    # 1M -> 1M: the two modes on the same GPU, every target: ids identical, weights / values within the tolerance;
    # then the oracle on a sample.  The share of solves repeated exactly is small but not zero.
    from multimesh_amd.device import Context

    pa, ca = synth.hex_mesh(101, seed=1)
    pb, _ = synth.hex_mesh(101, seed=7)
    fields = synth.vector_field(pa)
    vals, enc, w, nf = ctx.interpolate_hex8(pa, ca, pb, fields, want_operator=True)
    stats = ctx.last_locate_stats()
    ex = Context(0)
    try:
        vals_e, enc_e, w_e, nf_e = ex.interpolate_hex8(pa, ca, pb, fields, want_operator=True)
        assert ex.last_locate_stats()["redone_exact"] == 0
        assert nf == nf_e == 0
        assert np.array_equal(enc.numpy(), enc_e.numpy())
        assert close(w.numpy(), w_e.numpy()) and close(vals.numpy(), vals_e.numpy(), 8 * np.abs(fields).max())
    finally:
        ex.close()
    assert 0 < stats["redone_exact"] < 0.1 * len(pb)
    assert np.abs(w.numpy().sum(axis=1) - 1).max() < 1e-13
    pick = np.sort(np.random.default_rng(3).choice(len(pb), size=30_000, replace=False))
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb[pick], 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb[pick])
    assert np.array_equal(enc.numpy()[pick], enc_o) and close(w.numpy()[pick], w_o)


def test_legacy_symbol_opts_in_through_the_environment(golden):
    # MM_FP_MODE=tol: the process-wide context of the legacy symbols starts in MM_FP_TOL (a child process: the
    # variable is read when a context is created)
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from multimesh_amd import helpers
d = np.load(%r)
lib = helpers.load_lib()
n, k = d["nn"].shape
enc = np.zeros((n, 8), np.int64); w = np.zeros((n, 8))
nf = lib.triLinearInterpolator(k, n, np.ascontiguousarray(d["nn"]), np.ascontiguousarray(d["conn_reordered"]), enc,
                               np.ascontiguousarray(d["points_a"]), w, np.ascontiguousarray(d["points_b"]))
assert nf == int(d["nfailed"]) and np.array_equal(enc, d["enc"])
diff = np.abs(w - d["w"]).max()
assert 0 < diff <= 1e-12, diff     # the cheaper arithmetic really ran (some weight differs in its last bits)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "hex8_hard_k20.npz"))
    env = dict(os.environ, MM_FP_MODE="tol")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_metric_size_10M_tol_against_exact_on_every_target(ctx):
    # BASELINE's metric configuration (10,077,696 -> 10,077,696, what bench.py runs in MM_FP_TOL): the fused values-only
    # call and the operator call in both modes on the same GPU -- node ids identical on EVERY target, weights / values within
    # the tolerance, partition of unity, the trilinear field reproduced; then the oracle (cKDTree over all source centroids
    # + the C restatement) on a random sample.  Size-independent properties carry the full size, the oracle a sample.
    from multimesh_amd.device import Context

    pa, ca = synth.hex_mesh(216, seed=1)
    pb, _ = synth.hex_mesh(216, seed=7)
    fields = synth.vector_field(pa)[:1]
    d = [ctx.to_device(x) for x in (pa, ca, pb, fields)]
    vals, enc, w, nf = ctx.interpolate_hex8(*d, want_operator=True)
    stats = ctx.last_locate_stats()
    vals_only, nf_only = ctx.interpolate_hex8(*d)
    assert nf == nf_only == 0
    assert np.array_equal(vals_only.numpy(), vals.numpy())            # the two TOL calls agree bit for bit (deterministic)
    enc_t, w_t, vals_t = enc.numpy(), w.numpy(), vals.numpy()
    del enc, w, vals, vals_only
    ex = Context(0)
    try:
        de = [ex.to_device(x) for x in (pa, ca, pb, fields)]
        vals_e, enc_e, w_e, nf_e = ex.interpolate_hex8(*de, want_operator=True)
        assert nf_e == 0 and ex.last_locate_stats()["redone_exact"] == 0
        assert np.array_equal(enc_t, enc_e.numpy())                   # every one of the 10 M targets
        assert close(w_t, w_e.numpy()) and close(vals_t, vals_e.numpy(), 8 * np.abs(fields).max())
    finally:
        ex.close()
    assert 0 < stats["redone_exact"] < 0.05 * len(pb)
    assert np.abs(w_t.sum(axis=1) - 1).max() < 1e-13
    assert np.abs(vals_t[:, 0] - synth.field_linear(pb)).max() < 1e-7
    pick = np.sort(np.random.default_rng(3).choice(len(pb), size=20_000, replace=False))
    nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pb[pick], 20, workers=-1)
    enc_o, w_o, nf_o = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pb[pick])
    assert nf_o == 0 and np.array_equal(enc_t[pick], enc_o) and close(w_t[pick], w_o)
    assert close(vals_t[pick], O.gather(fields, enc_o, w_o), 8 * np.abs(fields).max())
