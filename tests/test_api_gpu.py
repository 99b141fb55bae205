"""The reference-named host API (multimesh_amd.api) and the sharded driver, on the GPU."""
import os
import socket

import numpy as np
import pytest

from multimesh_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _meshes(n_a=9, n_b=10):
    from multimesh_amd.mesh import HexMesh

    pa, ca = synth.hex_mesh(n_a, seed=1)
    pb, cb = synth.hex_mesh(n_b, seed=7)
    fields = {"VSV": synth.field_linear(pa), "VSH": synth.field_smooth(pa), "RHO": synth.field_xyz(pa)}
    return HexMesh(pa, ca, fields), HexMesh(pb, cb)


def _oracle_values(mesh_a, points, names, k):
    nn, _ = O.knn_ckdtree(O.centroid(mesh_a.connectivity, mesh_a.points), points, k)
    enc, w, nf = O.locate_hex8(nn, synth.reorder_hex8(mesh_a.connectivity), mesh_a.points, points)
    return O.gather(mesh_a.fields_matrix(names), enc, w), enc, w, nf


def test_interpolate_mesh_a_to_b_matches_reference_flow():
    from multimesh_amd import api

    a, b = _meshes()
    api.interpolate_mesh_a_to_b(a, b, params=["VSV", "VSH", "RHO"])          # reference cli.py:41-104
    truth, _, _, nf = _oracle_values(a, b.points, ["VSV", "VSH", "RHO"], 20)
    assert nf == 0
    for i, name in enumerate(["VSV", "VSH", "RHO"]):
        assert np.array_equal(b.get_nodal_field(name), truth[:, i])
    assert np.array_equal(a.get_element_centroid(), O.centroid(a.connectivity, a.points))


def test_interpolate_to_points_and_operator_split():
    from multimesh_amd import api

    a, _ = _meshes()
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.1, 1.1, size=(500, 3))                                # some outside -> zeros
    vals = api.interpolate_to_points(a, pts, ["VSV", "RHO"])                   # reference api.py:320-350 (k = 25)
    truth, enc_o, w_o, nf = _oracle_values(a, pts, ["VSV", "RHO"], 25)
    assert nf > 0 and np.array_equal(vals, truth)
    assert not vals[~w_o.any(axis=1)].any()                                     # failed points are zero
    # operator once, apply many times (reference stored_array split)
    enc, w, nfailed = api.interpolate_operator(a, pts, nelem_to_search=25)
    assert nfailed == nf and np.array_equal(enc, enc_o) and np.array_equal(w, w_o)
    assert np.array_equal(api.apply_operator(a, enc, w, ["VSV", "RHO"]), truth)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):                          # HDF5 PATHS need h5py; open objects do not
            api.gll_2_gll("a.h5", "b.h5")


def test_stored_operator_cache_round_trip(tmp_path):
    from multimesh_amd import api

    a, b = _meshes()
    first = api.interpolate_cached(a, b.points, ["VSV", "RHO"], stored_array=str(tmp_path / "op"))
    assert (tmp_path / "op" / "coeffs.npy").exists() and (tmp_path / "op" / "elements.npy").exists()
    a.attach_field("VSV", 2.0 * a.get_nodal_field("VSV"))                         # a new model iteration
    second = api.interpolate_cached(a, b.points, ["VSV", "RHO"], stored_array=str(tmp_path / "op"))
    assert np.array_equal(second[:, 1], first[:, 1]) and np.array_equal(second[:, 0], 2.0 * first[:, 0])
    truth, _, _, _ = _oracle_values(a, b.points, ["VSV", "RHO"], 20)
    assert np.array_equal(second, truth)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, tmpdir):
    import torch
    import torch.distributed as dist

    from multimesh_amd.distributed import HipShardInterpolator, interpolate_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # one GPU on this box: gloo carries the gather
    try:
        pa, ca = synth.hex_mesh(21, seed=1)
        pb, _ = synth.hex_mesh(23, seed=7)
        hip = HipShardInterpolator(pa, ca, synth.vector_field(pa), nelem_to_search=20, device_index=0)

        def local(shard):
            out, nf = hip(shard)
            return out.cpu(), nf

        vals, nfailed = interpolate_sharded(pb, local)
        np.save(os.path.join(tmpdir, f"v{rank}.npy"), vals.numpy())
        assert nfailed == 0
        # the GLL path shards the same way
        from multimesh_amd.distributed import HipShardGllInterpolator

        gp = synth.gll_mesh(6, 2, seed=3)
        gll = HipShardGllInterpolator(gp, 2, synth.field_linear(gp), nelem_to_search=20, device_index=0)

        def local_gll(shard):
            out, nm = gll(shard)
            return out.cpu(), nm

        gvals, gmiss = interpolate_sharded(pb[:5001], local_gll)
        np.save(os.path.join(tmpdir, f"g{rank}.npy"), gvals.numpy())
        assert gmiss == 0
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_and_agree_with_one(tmp_path):
    import torch.multiprocessing as mp

    from multimesh_amd.device import Context

    mp.spawn(_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    pa, ca = synth.hex_mesh(21, seed=1)
    pb, _ = synth.hex_mesh(23, seed=7)
    with Context(0) as ctx:
        single, nf = ctx.interpolate_hex8(pa, ca, pb, synth.vector_field(pa))
        single = single.numpy()
    assert nf == 0
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"v{r}.npy"), single)        # partition independent
    gp = synth.gll_mesh(6, 2, seed=3)
    with Context(0) as ctx:
        gsingle, gm = ctx.interpolate_gll(2, gp, pb[:5001], synth.field_linear(gp))
        gsingle = gsingle.numpy()
    assert gm == 0 and np.abs(gsingle[:, 0] - synth.field_linear(pb[:5001])).max() < 1e-11
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"g{r}.npy"), gsingle)


# ------------------------------------------------------------------------------- A11
@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_unique_points_equal_numpy(dim):
    # element-nodal GLL points: shared faces / edges / corners repeat (reference utils.py:484-488)
    from multimesh_amd.device import Context
    from multimesh_amd import synth
    pts = synth.gll_mesh(9 if dim == 3 else 40, 4, seed=3, dim=dim).reshape(-1, dim)
    rng = np.random.default_rng(0)
    pts = pts[rng.permutation(len(pts))]                    # arbitrary input order
    pts[::97, 0] = 0.0
    pts[::194, 0] = -0.0                                    # -0.0 == 0.0 like NumPy
    ctx = Context(0)
    uniq, inv = ctx.unique_points(pts)
    uniq, inv = uniq.numpy(), inv.numpy()
    ref_u, ref_inv = np.unique(pts, axis=0, return_inverse=True)
    assert uniq.shape == ref_u.shape and len(ref_u) < len(pts)
    assert np.array_equal(uniq, ref_u)                      # == : the sign of a zero is not compared
    assert np.array_equal(inv, ref_inv.reshape(-1))
    assert np.array_equal(uniq[inv], pts)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["general", "box_faces", "many_long_runs", "lattice", "x_differs_in_the_last_bits",
                                  "a_plane_with_noise_in_x"])
def test_unique_points_runs_of_equal_x(kind):
    # one sort by x, then the runs of equal x: short ones (copies of shared nodes) in place, the few long ones (the
    # faces of a box mesh) as a sub-sort, clouds that are mostly long runs by dim stable sorts -- every route against np.unique
    from multimesh_amd.device import Context
    rng = np.random.default_rng(5)
    n = 150_000
    pts = rng.uniform(size=(n, 3))
    if kind == "box_faces":
        pts[:9000, 0] = 0.0
        pts[9000:14000, 0] = 1.0
        pts[14000:14040, 0] = 0.5                           # a run just above the in-place limit
        pts[:14040, 1] = np.round(pts[:14040, 1], 2)        # ties in y inside the long runs: z decides
    elif kind == "many_long_runs":
        pts[:, 0] = np.round(pts[:, 0] * 5000) / 5000       # 5001 values of x, ~30 rows each: some runs longer than the limit
        pts[:60_000, 0] = np.round(pts[:60_000, 0] * 400) / 400   # ... and 401 values with 150 rows each
    elif kind == "x_differs_in_the_last_bits":
        # the main sort orders by the top 48 bits of x: rows whose x agree in those and differ below form runs that the
        # fix-up has to put in full (x, y, z) order -- short ones, and one longer than the in-place limit
        base = np.round(pts[:, 0] * 3000) / 3000 + 0.25
        pts[:, 0] = base * (1.0 + rng.integers(0, 200, size=n) * 2.0 ** -52)
        pts[:500, 0] = 0.625 * (1.0 + rng.integers(0, 4000, size=500) * 2.0 ** -52)
    elif kind == "a_plane_with_noise_in_x":
        pts[:, 0] = 3.0 * (1.0 + rng.integers(0, 1 << 15, size=n) * 2.0 ** -52)     # ONE run: the dim-sorts route
    elif kind == "lattice":
        g = np.arange(53, dtype=np.float64) / 52
        pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)[rng.permutation(53 ** 3)]
    pts = np.concatenate([pts, pts[rng.integers(0, len(pts), size=len(pts) // 3)]])   # exact duplicates of whole rows
    pts = np.ascontiguousarray(pts[rng.permutation(len(pts))])
    ctx = Context(0)
    uniq, inv = ctx.unique_points(pts)
    ref_u, ref_inv = np.unique(pts, axis=0, return_inverse=True)
    assert np.array_equal(uniq.numpy(), ref_u) and np.array_equal(inv.numpy(), ref_inv.reshape(-1))
    ctx.close()


def _first_occurrence_form(pts):
    """What mm_unique_points_any_order must return: np.unique's classes in the order of their first row."""
    n = len(pts)
    ru, rinv = np.unique(pts, axis=0, return_inverse=True)
    rinv = rinv.reshape(-1)
    first = np.full(len(ru), n, dtype=np.int64)
    np.minimum.at(first, rinv, np.arange(n))
    order = np.argsort(first)
    pos = np.empty(len(ru), dtype=np.int64)
    pos[order] = np.arange(len(ru))
    return pts[first[order]] + 0.0, pos[rinv]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gll3", "gll2", "lattice", "few_values", "signed_zeros", "one_row", "all_equal"])
def test_unique_points_any_order_is_the_same_collapse_without_the_sort(kind):
    # the order-free form (a hash table: reference interpolator.py:823 / :1079-1081 only ever scatter values back through the
    # inverse): the SAME classes as np.unique, rows in the order of their first occurrence, an inverse that rebuilds the input;
    # deterministic (two runs agree) whatever the order of the atomics
    from multimesh_amd.device import Context
    from multimesh_amd import synth
    rng = np.random.default_rng(11)
    if kind == "gll3":
        pts = synth.gll_mesh(12, 4, seed=3, dim=3).reshape(-1, 3)
    elif kind == "gll2":
        pts = synth.gll_mesh(60, 4, seed=4, dim=2).reshape(-1, 2)
    elif kind == "lattice":
        g = np.arange(40, dtype=np.float64) / 39
        pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
        pts = np.concatenate([pts, pts[rng.integers(0, len(pts), size=50_000)]])[rng.permutation(len(pts) + 50_000)]
    elif kind == "few_values":
        pts = rng.integers(-3, 4, size=(300_000, 3)).astype(np.float64)      # 343 classes: long chains on few slots
    elif kind == "signed_zeros":
        pts = rng.integers(-1, 2, size=(5000, 2)).astype(np.float64) * rng.choice([1.0, -0.0], size=(5000, 2))
    elif kind == "one_row":
        pts = np.array([[1.0, 2.0, 3.0]])
    else:
        pts = np.tile(np.array([[0.25, -7.0, 1e300]]), (10_000, 1))
    pts = np.ascontiguousarray(pts)
    ctx = Context(0)
    try:
        u, inv = ctx.unique_points(pts, ordered=False)
        u, inv = u.numpy(), inv.numpy()
        want_u, want_inv = _first_occurrence_form(pts)
        assert u.shape == want_u.shape and np.array_equal(u, want_u) and np.array_equal(inv, want_inv)
        assert np.array_equal(u[inv], pts) and not np.signbit(u[u == 0]).any()
        u2, inv2 = ctx.unique_points(pts, ordered=False)
        assert np.array_equal(u2.numpy(), u) and np.array_equal(inv2.numpy(), inv)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_unique_points_edge_cases():
    from multimesh_amd.device import Context
    ctx = Context(0)
    one = np.array([[1.0, 2.0, 3.0]])
    u, inv = ctx.unique_points(one)
    assert np.array_equal(u.numpy(), one) and inv.numpy().tolist() == [0]
    same = np.tile(one, (1000, 1))
    u, inv = ctx.unique_points(same)
    assert u.numpy().shape == (1, 3) and not inv.numpy().any()
    neg = np.array([[-1.0, 5.0], [-2.0, 7.0], [-1.0, -5.0], [3.0, 0.0], [-2.0, 7.0]])
    u, inv = ctx.unique_points(neg)
    ref_u, ref_inv = np.unique(neg, axis=0, return_inverse=True)
    assert np.array_equal(u.numpy(), ref_u) and np.array_equal(inv.numpy(), ref_inv.reshape(-1))
    ctx.close()


@pytest.mark.gpu
def test_gll_to_gll_on_arrays_reproduces_a_polynomial():
    # gll_2_gll's array core: unique targets, GLL locate + gather, scatter back with the inverse index
    from multimesh_amd import api, synth
    src = synth.gll_mesh(7, 4, seed=1, dim=3)
    tgt = synth.gll_mesh(6, 4, seed=7, dim=3)
    poly = lambda p: 1.0 + 2.0 * p[..., 0] - 3.0 * p[..., 1] * p[..., 2] + p[..., 0] ** 2
    mesh = api.GllMesh(src, 4, {"f": poly(src)})
    out = api.interpolate_gll_to_gll(mesh, tgt, ["f"])
    assert out.shape == (1,) + tgt.shape[:2]
    assert np.abs(out[0] - poly(tgt)).max() < 1e-9          # inside the order-4 space: exact to rounding


@pytest.mark.gpu
def test_hex8_to_gll_on_arrays_equals_the_pointwise_path():
    # exodus_2_gll's array core: unique GLL targets through the hex8 pipeline, scattered back
    from multimesh_amd import api, synth
    from multimesh_amd.mesh import HexMesh
    pa, ca = synth.hex_mesh(20, seed=1)
    mesh = HexMesh(pa, ca, {"f1": synth.field_linear(pa), "f2": synth.field_smooth(pa)})
    tgt = synth.gll_mesh(7, 4, seed=7, dim=3)
    out = api.interpolate_hex8_to_gll(mesh, tgt, ["f1", "f2"])
    assert out.shape == (2,) + tgt.shape[:2]
    flat = api.interpolate_to_points(mesh, tgt.reshape(-1, 3), ["f1", "f2"], nelem_to_search=20)
    assert np.array_equal(out.reshape(2, -1).T, flat)               # same bits as one call per point
    assert np.abs(out[0] - synth.field_linear(tgt)).max() < 1e-7    # trilinear field reproduced


@pytest.mark.gpu
@pytest.mark.parametrize("order,dim", [(4, 3), (2, 2)])
def test_gll_to_mesh_nodes_array_core_of_gll_2_exodus(order, dim):
    # reference interpolator.py:227-285 spelled out with the oracle: per-dimension np.mean centroids,
    # k-d tree, bounding-box acceptance loop, np.sum(gll_data[element, :, :] * coeffs, axis=1)
    from multimesh_amd import api

    gp = synth.gll_mesh(6 if dim == 3 else 9, order, seed=5, dim=dim)
    rng = np.random.default_rng(3)
    data = np.stack([synth.field_linear(gp), synth.field_smooth(gp.reshape(-1, dim)).reshape(gp.shape[:2]),
                     rng.normal(size=gp.shape[:2])], axis=1)                      # [E, C, P] like MODEL/data
    pts = rng.uniform(0.0, 1.0, size=(4000, dim))
    vals = api.interpolate_gll_to_nodes(gp, data, pts, shape_order=order, nelem_to_search=20)
    cen = np.stack([np.mean(gp[:, :, d], axis=1, dtype=np.float64) for d in range(dim)], axis=1)
    assert np.array_equal(api.find_gll_centroids(gp, dim), cen)
    nn, _ = O.knn_ckdtree(cen, pts, min(20, len(gp)))
    elem, coeffs, _ = O.locate_gll_v1(order, nn, gp, pts)
    want = np.stack([np.sum(data[elem[s], :, :] * coeffs[s], axis=1) for s in range(len(pts))])
    assert np.array_equal(vals, want)
    assert np.abs(vals[:, 0] - synth.field_linear(pts)).max() < 1e-11


@pytest.mark.gpu
def test_query_gll_model_array_core():
    # reference interpolator.py:60-139: tree over all GLL points, floor(index / P) -> element lists with
    # repeats, bounding-box loop, np.sum(original_data[elements] * coeffs, axis=2).  The copies of a
    # shared node are equidistant: the brute-force oracle orders them by index like the device.
    from multimesh_amd import api

    gp = synth.gll_mesh(5, 2, seed=8, dim=3)                                     # 64 elements x 27 nodes
    rng = np.random.default_rng(4)
    data = np.stack([synth.field_linear(gp), rng.normal(size=gp.shape[:2])], axis=1)   # [E, C, P]
    pts = rng.uniform(0.02, 0.98, size=(1500, 3))
    vals = api.query_gll_model(gp, data, pts, nelem_to_search=20)
    P = gp.shape[1]
    nearest = np.floor(O.knn_brute(gp.reshape(-1, 3), pts, 20) / P).astype(np.int64)
    assert (np.sort(nearest, axis=1)[:, 1:] == np.sort(nearest, axis=1)[:, :-1]).any()   # repeats do occur
    elem, coeffs, hard = O.locate_gll_v1(2, nearest, gp, pts)
    assert hard == 0
    want = np.sum(data[elem] * coeffs[:, None, :], axis=2)
    assert np.array_equal(vals, want)
    assert np.abs(vals[:, 0] - synth.field_linear(pts)).max() < 1e-11


# ------------------------------------------------------------------------------- section 8f-4: layers
def _layered_oracle(src, layer_a, fields, tgt, layer_b, layers, order, k, tol):
    """The reference's loop over layers (interpolator.py:1047-1082) spelled out with NumPy + the CPU oracle."""
    out = np.zeros((fields.shape[0],) + tgt.shape[:2])
    ops = {}
    for layer in layers:
        sm, tm = layer_a == layer, layer_b == layer
        nodes = tgt[tm]
        uniq, inv = np.unique(nodes.reshape(-1, nodes.shape[2]), return_inverse=True, axis=0)   # utils.py:506-510
        nn, _ = O.knn_ckdtree(src[sm].mean(axis=1), uniq, k)                                     # tree over the masked centroids
        elem, co, _ = O.locate_gll(order, nn, np.ascontiguousarray(src[sm]), uniq, tolerance=tol, snap_to_nearest=True)
        vals = O.gather_elem(np.ascontiguousarray(fields[:, sm]), elem, co)                      # [U, C]
        out[:, tm] = vals[inv.reshape(-1)].reshape(nodes.shape[0], nodes.shape[1], -1).transpose(2, 0, 1)
        ops[str(layer)] = (elem, co)
    return out, ops


@pytest.mark.gpu
def test_layered_gll_to_gll_equals_the_per_layer_loop(tmp_path):
    # a 3-layer synthetic "Earth": layers are slabs in z; source and target meshes of different resolution, so
    # that target points near a layer boundary have nearest source centroids in the WRONG layer -- which the
    # per-layer trees must never offer
    from multimesh_amd import api, synth

    order = 2
    src = synth.gll_mesh(10, order, seed=1)                      # 9^3 elements
    tgt = synth.gll_mesh(13, order, seed=7)                      # 12^3 elements
    layer_a = np.minimum((src.mean(axis=1)[:, 2] * 3).astype(int), 2)
    layer_b = np.minimum((tgt.mean(axis=1)[:, 2] * 3).astype(int), 2)
    fields = {"VSV": 1.0 + synth.field_smooth(src.reshape(-1, 3)).reshape(src.shape[:2]) ** 2 + layer_a[:, None],
              "RHO": synth.field_linear(src) + 10.0 * layer_a[:, None]}
    mesh = api.GllMesh(src, order, fields)
    fstack = np.stack([fields["VSV"], fields["RHO"]])
    want, ops = _layered_oracle(src, layer_a, fstack, tgt, layer_b, [2, 1, 0], order, 30, 1.05)
    store = str(tmp_path / "op")
    got = api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV", "RHO"], layers="all", stored_array=store)
    assert got.shape == want.shape and np.array_equal(got, want)
    # the stored per-layer operator (interp_info: coeffs/<layer>, elements/<layer>) equals the oracle's and is re-applied
    el, co = api.load_stored_layer_operator(store)
    for key, (e_o, c_o) in ops.items():
        assert np.array_equal(el[key], e_o) and np.array_equal(co[key], c_o)
    again = api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV", "RHO"], layers="all", stored_array=store)
    assert np.array_equal(again, want)
    # a subset of the layers: the other target elements keep what they had
    existing = np.full(want.shape, -7.0)
    part = api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV", "RHO"], layers=[1], existing=existing)
    assert np.array_equal(part[:, layer_b == 1], want[:, layer_b == 1]) and np.all(part[:, layer_b != 1] == -7.0)
    # every value comes from the target's own layer: RHO carries 10 * layer (points that stick out of their layer
    # are snapped to its nearest element, xi clipped to +-1.02 as in the reference: not the exact linear value)
    assert np.array_equal(np.round((got[1] - synth.field_linear(tgt)) / 10.0), np.broadcast_to(layer_b[:, None], got[1].shape))
    with pytest.raises(ValueError):
        api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV"], layers=[5])


@pytest.mark.gpu
def test_layer_presets_and_the_bounding_box_acceptance_of_the_older_drivers():
    # a 5-layer "Earth" in z with a fluid layer: the presets are resolved on the SOURCE mesh (reference
    # interpolator.py:1019) exactly as utils._assess_layers does (:413-436), and acceptance="bbox" is the loop of
    # gll_2_gll_layered / gll_2_gll_layered_multi (fill_value_array -> _check_if_inside_element)
    from multimesh_amd import api, synth

    order = 2
    src = synth.gll_mesh(11, order, seed=1)
    tgt = synth.gll_mesh(9, order, seed=7)
    layer_a = np.minimum((src.mean(axis=1)[:, 2] * 5).astype(int), 4) + 1        # layers 1..5, 5 = top
    layer_b = np.minimum((tgt.mean(axis=1)[:, 2] * 5).astype(int), 4) + 1
    fluid_a = (layer_a == 2).astype(float)                                       # the "outer core"
    fields = np.stack([1.0 + synth.field_smooth(src.reshape(-1, 3)).reshape(src.shape[:2]) ** 2 + layer_a[:, None]])
    mesh = api.GllMesh(src, order, {"VSV": fields[0]})
    moho_idx = 1
    # reference utils.py:396, 411-436 spelled out
    mesh_layers = np.sort(np.unique(layer_a))[::-1].astype(int)
    o_core = np.where(mesh_layers == layer_a[np.where(fluid_a == 1)[0][0]])[0][0]
    presets = {"crust": mesh_layers[:moho_idx], "mantle": mesh_layers[moho_idx:o_core], "core": mesh_layers[o_core:],
               "nocore": mesh_layers[:o_core], "all": mesh_layers}
    for name, want_layers in presets.items():
        assert api.assess_layers(layer_a, name, fluid=fluid_a, moho_idx=moho_idx) == [int(x) for x in want_layers]
    existing = np.full((1,) + tgt.shape[:2], -3.0)
    for name in ("nocore", "mantle"):
        want, _ = _layered_oracle(src, layer_a, fields, tgt, layer_b, list(presets[name]), order, 30, 1.05)
        got = api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV"], layers=name, fluid_a=fluid_a,
                                                 moho_idx=moho_idx, existing=existing)
        inside = np.isin(layer_b, presets[name])
        assert np.array_equal(got[:, inside], want[:, inside]) and np.all(got[:, ~inside] == -3.0)
    with pytest.raises(ValueError, match="moho_idx"):
        api.assess_layers(layer_a, "crust", fluid=fluid_a)
    with pytest.raises(ValueError, match="fluid"):
        api.assess_layers(layer_a, "core")
    # the older drivers' acceptance loop, per layer (reference interpolator.py:1516-1538)
    store = None
    got = api.interpolate_gll_to_gll_layered(mesh, layer_a, tgt, layer_b, ["VSV"], layers="nocore", fluid_a=fluid_a,
                                             moho_idx=moho_idx, nelem_to_search=20, acceptance="bbox", stored_array=store)
    want = np.zeros_like(got)
    for layer in presets["nocore"]:
        sm, tm = layer_a == layer, layer_b == layer
        nodes = tgt[tm]
        uniq, inv = np.unique(nodes.reshape(-1, 3), return_inverse=True, axis=0)
        nn, _ = O.knn_ckdtree(src[sm].mean(axis=1), uniq, 20)
        elem, co, _ = O.locate_gll_v1(order, nn, np.ascontiguousarray(src[sm]), uniq)
        vals = O.gather_elem(np.ascontiguousarray(fields[:, sm]), elem, co)
        want[:, tm] = vals[inv.reshape(-1)].reshape(nodes.shape[0], nodes.shape[1], -1).transpose(2, 0, 1)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_fluid_solid_fix_equals_the_reference_statements():
    from multimesh_amd import api

    rng = np.random.default_rng(3)
    E, C, P = 500, 3, 27
    values = rng.uniform(1.0, 2.0, size=(E, C, P))
    new_values = rng.uniform(5.0, 6.0, size=(E, C, P))          # what the target mesh held before (interpolator.py:690)
    solid_elements = rng.random(E) > 0.3
    values[rng.choice(E, 60, replace=False), 1, rng.integers(0, P, 60)] = 0.0     # zero VSV here and there
    parameters = ["RHO", "VSV", "VPV"]
    got = api.fix_fluid_solid(values, new_values, solid_elements, parameters)
    # reference interpolator.py:829-841, verbatim on copies
    want = values.copy()
    want[~solid_elements] = new_values[~solid_elements]
    vs_index = parameters.index("VSV")
    zero_vs = np.where(want[:, vs_index, :] == 0.0)
    for _i, elem in enumerate(np.unique(zero_vs[0])):
        if solid_elements[elem]:
            want[elem, :, :] = new_values[elem, :, :]
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------- section 8f-2: file-level drivers
def _gll_model(gp, data, params, fluid=None, layer=None):
    """A Salvus GLL model in the HDF5 layout, held in memory (h5py is not in this image)."""
    from multimesh_amd import io as mio

    h = mio.MemoryH5()
    h.create_dataset("MODEL/coordinates", data=gp)
    mio.set_dimension_labels(h.create_dataset("MODEL/data", data=data), list(params))
    nelem = gp.shape[0]
    ed = np.stack([np.zeros(nelem) if fluid is None else fluid, np.zeros(nelem) if layer is None else layer], axis=1)
    h.create_dataset("MODEL/element_data", data=ed).attrs["DIMENSION_LABELS"] = np.array([b"element", b"[ fluid | layer ]"])
    return h


@pytest.mark.gpu
def test_exodus_2_gll_from_an_exodus_file_into_a_gll_model(tmp_path):
    # reference interpolator.py:142-224: exodus nodal fields -> every GLL point of the model, MODEL/data rewritten
    from multimesh_amd import api, io as mio

    pa, ca = synth.hex_mesh(14, seed=1)
    fields = {"VP": synth.field_linear(pa), "RHO": synth.field_smooth(pa), "unused": pa[:, 0]}
    fn = str(tmp_path / "coarse.e")
    mio.write_exodus_classic(fn, pa, ca, fields)
    gp = synth.gll_mesh(5, 2, seed=7, dim=3)
    model = _gll_model(gp, np.full((gp.shape[0], 5, gp.shape[1]), -1.0), ["A", "B", "C", "D", "E"])
    api.exodus_2_gll(fn, model, gll_order=2, parameters=["VP", "RHO"], nelem_to_search=20)
    out = model["MODEL/data"][()]
    assert out.shape == (gp.shape[0], 2, gp.shape[1]) and mio.dimension_labels(model["MODEL/data"]) == ["VP", "RHO"]
    mesh_a = mio.Exodus(fn)
    want, _, _, nf = _oracle_values(type("M", (), {"connectivity": mesh_a.connectivity, "points": mesh_a.points,
                                                    "fields_matrix": lambda self, n: np.stack([fields[x] for x in n])})(),
                                    gp.reshape(-1, 3), ["VP", "RHO"], 20)
    assert nf == 0 and np.array_equal(out.transpose(0, 2, 1).reshape(-1, 2), want)
    assert np.abs(out[:, 0, :] - synth.field_linear(gp)).max() < 1e-7


@pytest.mark.gpu
def test_gll_2_exodus_attaches_the_models_parameters_to_the_exodus_nodes(tmp_path):
    # reference interpolator.py:227-285
    from multimesh_amd import api, io as mio

    gp = synth.gll_mesh(6, 4, seed=5, dim=3)
    rng = np.random.default_rng(3)
    data = np.stack([synth.field_linear(gp), rng.normal(size=gp.shape[:2])], axis=1)
    model = _gll_model(gp, data, ["VS", "QMU"])
    pb, cb = synth.hex_mesh(9, seed=7)
    fn = str(tmp_path / "fine.e")
    mio.write_exodus_classic(fn, pb, cb, {"QMU": np.zeros(len(pb)), "VS": np.zeros(len(pb)), "other": np.ones(len(pb))})
    api.gll_2_exodus(model, fn, nelem_to_search=20)
    e = mio.Exodus(fn)
    want = api.interpolate_gll_to_nodes(gp, data, pb, shape_order=4, nelem_to_search=20)   # (oracle-checked above)
    assert np.array_equal(e.get_nodal_field("VS"), want[:, 0]) and np.array_equal(e.get_nodal_field("QMU"), want[:, 1])
    assert np.array_equal(e.get_nodal_field("other"), np.ones(len(pb)))
    assert np.abs(e.get_nodal_field("VS") - synth.field_linear(pb)).max() < 1e-11


@pytest.mark.gpu
def test_gll_2_gll_between_two_models_with_the_stored_operator(tmp_path):
    # reference interpolator.py:621-852, spelled out with NumPy + the CPU oracle
    from multimesh_amd import api, io as mio

    src = synth.gll_mesh(5, 2, seed=8, dim=3)
    rng = np.random.default_rng(6)
    vs = synth.field_linear(src)
    vs[3] = 0.0                                                          # a fluid source element: VS = 0
    data = np.stack([rng.normal(size=src.shape[:2]), vs], axis=1)         # parameters RHO, VS
    from_model = _gll_model(src, data, ["RHO", "VS"])
    tgt = synth.gll_mesh(6, 2, seed=9, dim=3)
    fluid = (np.arange(tgt.shape[0]) % 7 == 0) * 1.0
    previous = rng.normal(size=(tgt.shape[0], 2, tgt.shape[1]))
    to_model = _gll_model(tgt, previous, ["RHO", "VS"], fluid=fluid)
    store = str(tmp_path / "operator")
    api.gll_2_gll(from_model, to_model, nelem_to_search=20, stored_array=store)
    got = to_model["MODEL/data"][()]

    P = src.shape[1]
    uniq, recon = np.unique(tgt.reshape(-1, 3), return_inverse=True, axis=0)
    nearest = np.floor(O.knn_brute(src.reshape(-1, 3), uniq, 20) / P).astype(np.int64)
    elem, coeffs, _ = O.locate_gll_v1(2, nearest, src, uniq)
    values = np.sum(data[elem] * coeffs[:, None, :], axis=2)[recon.reshape(-1), :].reshape(tgt.shape[0], tgt.shape[1], 2).swapaxes(1, 2)
    solid = ~fluid.astype(bool)
    values[~solid] = previous[~solid]
    touched = np.unique(np.where(values[:, 1, :] == 0.0)[0])
    assert len(touched) > 0                                              # the fix-up has something to fix
    for e in touched:
        if solid[e]:
            values[e] = previous[e]
    assert np.array_equal(got, values) and mio.dimension_labels(to_model["MODEL/data"]) == ["RHO", "VS"]

    # the operator files: elements.npy [U], coeffs.npy [1, P, U]; a second run re-applies them, also when they
    # hold the reference's nparam identical copies
    assert np.array_equal(np.load(os.path.join(store, "elements.npy")), elem)
    assert np.load(os.path.join(store, "coeffs.npy")).shape == (1, P, len(uniq))
    again = _gll_model(tgt, previous, ["RHO", "VS"], fluid=fluid)
    api.gll_2_gll(from_model, again, nelem_to_search=1, stored_array=store)   # (k is not used: nothing is located)
    assert np.array_equal(again["MODEL/data"][()], values)
    np.save(os.path.join(store, "coeffs.npy"), np.repeat(np.load(os.path.join(store, "coeffs.npy")), 2, axis=0))
    third = _gll_model(tgt, previous, ["RHO", "VS"], fluid=fluid)
    api.gll_2_gll(from_model, third, stored_array=store)
    assert np.array_equal(third["MODEL/data"][()], values)
    # gradient=True: no fluid / solid fix-up
    grad = _gll_model(tgt, previous, ["RHO", "VS"], fluid=fluid)
    api.gll_2_gll(from_model, grad, stored_array=store, gradient=True)
    raw = np.sum(data[elem] * coeffs[:, None, :], axis=2)[recon.reshape(-1), :].reshape(tgt.shape[0], tgt.shape[1], 2).swapaxes(1, 2)
    assert np.array_equal(grad["MODEL/data"][()], raw)


@pytest.mark.gpu
def test_query_model_and_layered_driver_on_model_files(tmp_path):
    from multimesh_amd import api, io as mio

    # query_model: (lat, lon, depth) -> Cartesian -> the array core; a small model around a patch of the Earth
    gp = synth.gll_mesh(4, 2, seed=8, dim=3)
    r_earth = 6371000.0
    centre = np.array([r_earth - 50_000.0, 0.0, 0.0])
    gp_earth = centre + (gp - 0.5) * 80_000.0
    data = np.stack([synth.field_linear(gp), np.ones(gp.shape[:2])], axis=1)
    model = _gll_model(gp_earth, data, ["VP", "ONE"])
    rng = np.random.default_rng(2)
    lld = np.stack([rng.uniform(-0.2, 0.2, 50), rng.uniform(-0.2, 0.2, 50), rng.uniform(30_000, 70_000, 50)], axis=1)
    vals = api.query_model(lld, model, nelem_to_search=20)
    want = api.query_gll_model(gp_earth, data, api.latlondepth_to_xyz(lld), 20)
    assert vals.shape == (50, 2) and np.array_equal(vals, want) and np.abs(vals[:, 1] - 1.0).max() < 1e-12

    # gll_2_gll_layered_multi_two: the layer field of both files drives the per-layer passes; fields are attached
    src = synth.gll_mesh(5, 2, seed=1, dim=3)
    tgt = synth.gll_mesh(6, 2, seed=7, dim=3)
    layer_a = (src.mean(axis=1)[:, 2] > 0.5) * 1.0
    layer_b = (tgt.mean(axis=1)[:, 2] > 0.5) * 1.0
    fields = np.stack([synth.field_linear(src), synth.field_smooth(src.reshape(-1, 3)).reshape(src.shape[:2])], axis=1)
    from_model = _gll_model(src, fields, ["VP", "VS"], layer=layer_a)
    before = rng.normal(size=(tgt.shape[0], 2, tgt.shape[1]))
    to_model = _gll_model(tgt, before, ["VP", "VS"], layer=layer_b)
    api.gll_2_gll_layered_multi_two(from_model, to_model, layers=[1], nelem_to_search=20, parameters="all")
    got = to_model["MODEL/data"][()]
    mesh_a = api.GllMesh(src, 2, {"VP": fields[:, 0], "VS": fields[:, 1]})
    want = api.interpolate_gll_to_gll_layered(mesh_a, layer_a, tgt, layer_b, ["VP", "VS"], layers=[1], nelem_to_search=20,
                                              existing=before.transpose(1, 0, 2))
    assert np.array_equal(got, want.transpose(1, 0, 2))
    assert np.array_equal(got[layer_b == 0], before[layer_b == 0])        # other layers keep their values
    assert not np.array_equal(got[layer_b == 1], before[layer_b == 1])

    # the two older drivers (reference api.py:158-274): bounding-box acceptance per layer; gll_2_gll_layered zeroes
    # what it does not interpolate (interpolator.py:421), gll_2_gll_layered_multi keeps it (:606); a preset resolved
    # from the source file's fluid flag ("nocore" = layers above the fluid one: here layer 1 above fluid layer 0)
    fluid_a = 1.0 - layer_a
    from_model = _gll_model(src, fields, ["VP", "VS"], fluid=fluid_a, layer=layer_a)
    want_bbox = api.interpolate_gll_to_gll_layered(mesh_a, layer_a, tgt, layer_b, ["VP", "VS"], layers=[1], nelem_to_search=20,
                                                   acceptance="bbox")
    to_old = _gll_model(tgt, before, ["VP", "VS"], layer=layer_b)
    api.gll_2_gll_layered(from_model, to_old, layers="nocore", parameters=["VP", "VS"])
    assert np.array_equal(to_old["MODEL/data"][()], want_bbox.transpose(1, 0, 2))
    assert not to_old["MODEL/data"][()][layer_b == 0].any()
    to_multi = _gll_model(tgt, before, ["VP", "VS"], layer=layer_b)
    api.gll_2_gll_layered_multi(from_model, to_multi, parameters=["VP", "VS"], threads=4)    # layers="nocore" by default
    got_multi = to_multi["MODEL/data"][()]
    assert np.array_equal(got_multi[layer_b == 1], want_bbox.transpose(1, 0, 2)[layer_b == 1])
    assert np.array_equal(got_multi[layer_b == 0], before[layer_b == 0])
