"""Drop-in for the interpolate-between-meshes entry points of reference ``multi_mesh/api.py`` and
the hot-path command of ``multi_mesh/scripts/cli.py``.

Same function names, argument meaning and return conventions as the reference; mesh arguments are
:class:`multimesh_amd.mesh.HexMesh` array bundles because the reference's mesh readers (pyexodus,
h5py, salvus) are not part of the hot path.  Every function here ends in
HIP kernels through the C ABI of ``multi_mesh_hip.so``; there is no CPU fallback.

Covered (hex8, the path the reference implements in its own C):
  * :func:`interpolate_mesh_a_to_b`  -- reference scripts/cli.py:35-104
  * :func:`interpolate_to_points`    -- reference api.py:320-350 / interpolator.py:931-977
  * :func:`interpolate_to_mesh`      -- reference api.py:353-393
GLL paths (the reference's salvus.fem numerics restated): the array cores
:func:`interpolate_gll_to_points`, :func:`interpolate_gll_to_gll`, :func:`interpolate_gll_to_gll_layered`,
:func:`interpolate_hex8_to_gll`, :func:`interpolate_gll_to_nodes`, :func:`query_gll_model`, and over them the
file-level drivers under the reference's names -- :func:`query_model`, :func:`exodus_2_gll`,
:func:`gll_2_exodus`, :func:`gll_2_gll`, :func:`gll_2_gll_layered_multi_two` -- reading and writing meshes
through :mod:`multimesh_amd.io`.
"""
from __future__ import annotations

import os
import time

import numpy as np

from .device import default_context
from .helpers import check
from .mesh import HexMesh

TTI_PARAMS = ["VSH", "VSV", "VPV", "VPH", "RHO", "ETA", "QKAPPA", "QMU"]  # reference cli.py:58-59


def _report(start):
    # the reference prints wall-clock around every API call (api.py:39-57)
    runtime = time.time() - start
    if runtime >= 60:
        print(f"Finished in time: {runtime / 60} minutes")
    else:
        print(f"Finished in time: {runtime} seconds")


def latlondepth_to_xyz(latlondepth):
    """reference utils.py:526-542 (r_earth = 6371000 m, geocentric latitude)."""
    latlondepth = np.asarray(latlondepth, dtype=np.float64)
    r = 6371000.0 - latlondepth[:, 2]
    colat = np.deg2rad(90.0 - latlondepth[:, 0])
    lon = np.deg2rad(latlondepth[:, 1])
    return np.array([r * np.sin(colat) * np.cos(lon), r * np.sin(colat) * np.sin(lon), r * np.cos(colat)]).T


def interpolate_operator(mesh_a: HexMesh, points, nelem_to_search=20, context=None):
    """``(enclosing_elem_node_indices int64[N,8], weights f64[N,8], nfailed)`` for arbitrary points:
    centroid -> kNN -> locate, i.e. reference cli.py:62-95 without the field loop.  This is the
    persistable operator of the reference's ``stored_array`` split."""
    ctx = context or default_context()
    points = np.ascontiguousarray(points, dtype=np.float64)
    field = np.zeros((1, mesh_a.npoint))
    _, enc, w, nfailed = ctx.interpolate_hex8_host(mesh_a.points, mesh_a.connectivity, points, field,
                                                   nelem_to_search=nelem_to_search, want_operator=True)
    return enc, w, nfailed


def apply_operator(mesh_a: HexMesh, enclosing_elem_node_indices, weights, params, context=None):
    """``np.sum(param_a[enc] * weights, axis=1)`` per parameter (reference cli.py:98-100) -> f64[N,C]."""
    ctx = context or default_context()
    return ctx.gather(mesh_a.fields_matrix(params), enclosing_elem_node_indices, weights).numpy()


def load_stored_operator(stored_array):
    """The reference's operator cache (interpolator.py:724-740): ``elements.npy`` + ``coeffs.npy`` in
    the ``stored_array`` directory.  Returns ``(elements, coeffs)`` or ``None`` when not (fully) there.
    For the hex8 path ``elements`` holds the 8 node ids per point (``enclosing_elem_node_indices``)."""
    import os

    if not stored_array:
        return None
    e_path, c_path = os.path.join(stored_array, "elements.npy"), os.path.join(stored_array, "coeffs.npy")
    if not (os.path.exists(e_path) and os.path.exists(c_path)):
        return None
    coeffs = np.load(c_path, allow_pickle=True)
    elements = np.load(e_path, allow_pickle=True)
    assert not np.isnan(coeffs).any(), "Stored coeffs matrix has NaNs"          # interpolator.py:735-740
    return elements, coeffs


def save_stored_operator(stored_array, elements, coeffs):
    """reference interpolator.py:797-810"""
    import os

    if not os.path.exists(stored_array):
        os.makedirs(stored_array)
    print("Will save matrices for later usage")
    np.save(os.path.join(stored_array, "elements.npy"), elements, allow_pickle=True)
    np.save(os.path.join(stored_array, "coeffs.npy"), coeffs, allow_pickle=True)


def interpolate_cached(mesh_a: HexMesh, points, params, stored_array=None, nelem_to_search=20, context=None):
    """hex8 interpolation with the reference's ``stored_array`` split: the first
    call builds and stores the operator, later calls skip kNN + locate and run only the HBM-bound
    gather.  Returns f64[N, len(params)]."""
    cached = load_stored_operator(stored_array)
    if cached is None:
        enc, w, _ = interpolate_operator(mesh_a, points, nelem_to_search, context)
        if stored_array:
            save_stored_operator(stored_array, enc, w)
    else:
        print("Matrix was already stored. Will use that one")
        enc, w = cached
    return apply_operator(mesh_a, enc, w, params, context)


def interpolate_mesh_a_to_b(mesh_a: HexMesh, mesh_b: HexMesh, params=("TTI",), context=None):
    """Interpolates values from mesh A onto the nodes of mesh B (reference cli.py:41-104).

    Attaches every parameter to ``mesh_b`` and, like the reference, asserts that no point failed."""
    params = list(params)
    if params and params[0] == "TTI":
        params = list(TTI_PARAMS)
    ctx = context or default_context()
    nelem_to_search = 20  # reference cli.py:69
    values, nfailed = ctx.interpolate_hex8_host(mesh_a.points, mesh_a.connectivity, mesh_b.points,
                                                mesh_a.fields_matrix(params), nelem_to_search=nelem_to_search)
    for i, param in enumerate(params):
        mesh_b.attach_field(param, values[:, i])
    assert nfailed == 0, f"{nfailed} points could not be interpolated."
    return mesh_b


def interpolate_to_points(mesh, points, params_to_interp, make_spherical=False, geocentric=False,
                          nelem_to_search=25, context=None):
    """Maps values from a mesh to predefined points, xyz or geocentric latlondepth
    (reference api.py:320-350).  Returns f64[npoints, nparams]; points that are not found get zero
    (reference interpolator.py:963-977)."""
    if make_spherical:
        raise NotImplementedError("map_to_sphere (reference interpolator.py:1085-1144) is Earth-specific "
                                  "pre-processing outside the hot path (SURVEY.md §2 #15)")
    if geocentric:
        points = latlondepth_to_xyz(points)
    ctx = context or default_context()
    points = np.ascontiguousarray(points, dtype=np.float64)
    vals, nfailed = ctx.interpolate_hex8_host(mesh.points, mesh.connectivity, points,
                                              mesh.fields_matrix(params_to_interp), nelem_to_search=nelem_to_search)
    if nfailed > 0:
        print(nfailed, "points could not find an enclosing element. These points will be set to zero. "
                       "Please check your domain or the interpolation tuning parameters")
    return vals


def interpolate_to_mesh(old_mesh, new_mesh, params_to_interp=("VSV", "VSH", "VPV", "VPH"), context=None):
    """Interpolate ``params_to_interp`` from old_mesh onto the nodes of new_mesh
    (reference api.py:353-393, minus the Earth-specific sphere mapping).  Values that are not found
    are given zero."""
    start = time.time()
    vals = interpolate_to_points(old_mesh, new_mesh.points, list(params_to_interp), context=context)
    for i, param in enumerate(params_to_interp):
        new_mesh.attach_field(param, vals[:, i])
    _report(start)
    return new_mesh


class GllMesh:
    """Element-nodal GLL mesh bundle: what the reference reads from a Salvus mesh for the GLL path
    (``mesh.points[mesh.connectivity]``, ``mesh.shape_order``, ``mesh.element_nodal_fields``;
    interpolator.py:954-976)."""

    def __init__(self, gll_points, shape_order, element_nodal_fields=None):
        self.gll_points = np.ascontiguousarray(gll_points, dtype=np.float64)   # [E, P, dim]
        self.shape_order = int(shape_order)
        self.element_nodal_fields = {k: np.ascontiguousarray(v, dtype=np.float64)
                                     for k, v in (element_nodal_fields or {}).items()}

    @property
    def nelem(self):
        return self.gll_points.shape[0]

    def get_element_centroid(self):
        # the reference takes the mean of the control nodes (salvus_mesh_reader.py:99-100)
        return self.gll_points.mean(axis=1)


def get_element_weights(gll_points, shape_order, centroid_tree, points, nelem_to_search=25, tolerance=1.05,
                        snap_to_nearest=False, context=None):
    """Enclosing element and interpolation coefficients of every point
    (reference interpolator.py:1147-1255).  ``centroid_tree``: a :class:`multimesh_amd.device.KnnIndex`
    over the element centroids, or the centroid array itself.  Returns ``(elems int64[N] with -1 for
    "not found", coeffs f64[N, P])``."""
    ctx = context or default_context()
    tree = centroid_tree if hasattr(centroid_tree, "query") else ctx.knn_build(centroid_tree)
    nn = tree.query(points, nelem_to_search)
    elem, coeffs, _ = ctx.locate_gll(shape_order, nn, gll_points, points, tolerance, snap_to_nearest)
    return elem.numpy(), coeffs.numpy()


def check_if_inside_element(gll_model, nearest_elements, points, shape_order, context=None):
    """Array form of the reference's ``_check_if_inside_element`` (interpolator.py:1409-1473, called
    per point by gll_2_exodus and the layered drivers): bounding-box pre-test, acceptance at
    |xi| <= 1.04, best-candidate fallback.  ``gll_model`` f64[E, P, dim], ``nearest_elements``
    int64[N, k], ``points`` f64[N, dim] -> (element int64[N], coefficients f64[N, P])."""
    ctx = context or default_context()
    elem, coeffs, _ = ctx.locate_gll_bbox(shape_order, nearest_elements, gll_model, points)
    return elem.numpy(), coeffs.numpy()


def interpolate_gll_to_points(mesh: GllMesh, points, params_to_interp, nelem_to_search=25, tolerance=1.05,
                              context=None):
    """The GLL form of ``interpolate_to_points`` (reference interpolator.py:931-977): centroid tree,
    element weights, then ``np.sum(coeffs * field[elem], axis=1)`` per parameter -> f64[N, C]."""
    ctx = context or default_context()
    points = np.ascontiguousarray(points, dtype=np.float64)
    fields = np.stack([mesh.element_nodal_fields[p] for p in params_to_interp])
    vals, num_failed = ctx.interpolate_gll(mesh.shape_order, mesh.gll_points, points, fields,
                                           nelem_to_search=nelem_to_search, tolerance=tolerance)
    if num_failed > 0:
        print(num_failed, "points could not find an enclosing element. These points will be set to zero. "
                          "Please check your domain or the interpolation tuning parameters")
    return vals.numpy()


def get_unique_points(points, context=None):
    """Array form of the reference's ``utils.get_unique_points`` (utils.py:484-488):
    ``np.unique(points.reshape(-1, dim), axis=0, return_inverse=True)`` on the device.
    ``points`` f64[E, P, dim] (element-nodal) or f64[N, dim] -> (unique f64[U, dim], inverse int64[N])."""
    ctx = context or default_context()
    pts = np.ascontiguousarray(points, dtype=np.float64)
    uniq, inv = ctx.unique_points(pts.reshape(-1, pts.shape[-1]))
    return uniq.numpy(), inv.numpy()


def interpolate_gll_to_gll(mesh_a: GllMesh, target_gll_points, params_to_interp, nelem_to_search=20,
                           tolerance=1.05, context=None):
    """The array core of ``gll_2_gll`` (reference interpolator.py:700-830): the target mesh's
    element-nodal points are reduced to their unique set (shared faces/edges/corners repeat),
    interpolated once each, and scattered back with the inverse index (``values[recon]``,
    interpolator.py:823).  ``target_gll_points`` f64[E_t, P_t, dim] -> f64[C, E_t, P_t]."""
    ctx = context or default_context()
    tgt = np.ascontiguousarray(target_gll_points, dtype=np.float64)
    # (the unique rows are only interpolated and scattered back: their order never reaches the result)
    uniq, inv = ctx.unique_points(tgt.reshape(-1, tgt.shape[-1]), ordered=False)
    fields = np.stack([mesh_a.element_nodal_fields[p] for p in params_to_interp])
    vals, num_failed = ctx.interpolate_gll(mesh_a.shape_order, mesh_a.gll_points, uniq, fields,
                                           nelem_to_search=nelem_to_search, tolerance=tolerance)
    if num_failed > 0:
        print(num_failed, "points could not find an enclosing element. These points will be set to zero. "
                          "Please check your domain or the interpolation tuning parameters")
    vals = vals.numpy()                                             # [U, C]
    return np.ascontiguousarray(vals[inv.numpy()].T).reshape(len(params_to_interp), tgt.shape[0], tgt.shape[1])


def interpolate_hex8_to_gll(mesh_a: HexMesh, target_gll_points, params, nelem_to_search=20, context=None,
                            return_nfailed=False):
    """The array core of ``exodus_2_gll`` (reference cli.py:128-257, interpolator.py:60-150): the
    reference runs its hex8 path once per GLL slot (125 times at order 4) over points that repeat on
    shared faces, edges and corners; here the target mesh's element-nodal points are reduced to
    their unique set on the device, interpolated once each through the hex8 pipeline, and scattered
    back with the inverse index.  ``target_gll_points`` f64[E_t, P_t, 3] -> f64[C, E_t, P_t]; points
    that are not found get zero."""
    ctx = context or default_context()
    tgt = np.ascontiguousarray(target_gll_points, dtype=np.float64)
    uniq, inv = ctx.unique_points(tgt.reshape(-1, tgt.shape[-1]), ordered=False)
    vals, nfailed = ctx.interpolate_hex8(mesh_a.points, mesh_a.connectivity, uniq, mesh_a.fields_matrix(list(params)),
                                         nelem_to_search=nelem_to_search)
    if nfailed > 0:
        print(nfailed, "points could not find an enclosing element. These points will be set to zero. "
                       "Please check your domain or the interpolation tuning parameters")
    vals = vals.numpy()                                              # [U, C]
    out = np.ascontiguousarray(vals[inv.numpy()].T).reshape(len(list(params)), tgt.shape[0], tgt.shape[1])
    return (out, nfailed) if return_nfailed else out


def assess_layers(layer_ids, layers, fluid=None, moho_idx=None):
    """The reference's ``utils._assess_layers`` (utils.py:382-440) on arrays: ``layer_ids`` = the mesh's
    ``layer`` elemental field; ``layers`` = "all", a list of layer numbers (which must lie within the mesh's
    own), one layer number, or an Earth preset.  The mesh's layers are sorted in DESCENDING order, "outwards from
    the core" reversed, as the reference sorts them (:396); with ``o_core_idx`` = the place in that order of the
    layer of the first fluid element (:426-429):

        "crust"  -> layers[:moho_idx]            "mantle" -> layers[moho_idx:o_core_idx]
        "core"   -> layers[o_core_idx:]          "nocore" -> layers[:o_core_idx]

    ``fluid``: the mesh's ``fluid`` elemental field (needed by "mantle", "core", "nocore"); ``moho_idx``: the
    mesh's global string of that name (``mesh.global_strings["moho_idx"]``, needed by "crust" and "mantle")."""
    mesh_layers = np.sort(np.unique(np.asarray(layer_ids)))[::-1].astype(int)
    if isinstance(layers, (list, tuple, np.ndarray)):
        layers = [int(x) for x in np.atleast_1d(layers)]
        if max(layers) > mesh_layers.max() or min(layers) < mesh_layers.min():
            raise ValueError("Requested layers not in mesh")
        return layers
    if isinstance(layers, (int, np.integer)):
        if int(layers) not in mesh_layers:
            raise ValueError("Requested layer not in mesh")
        return [int(layers)]
    available_layers = ["all", "crust", "mantle", "core", "nocore"]
    if not isinstance(layers, str):
        raise ValueError(f"Input for layers needs to be a list of one of: {available_layers}")
    if layers == "all":
        return [int(x) for x in mesh_layers]
    if layers not in available_layers:
        raise ValueError(f"Only allowed string layer inputs are: {available_layers}")
    if layers in ("crust", "mantle"):
        if moho_idx is None:
            raise ValueError(f'layers="{layers}" needs the mesh\'s moho_idx (global string of the Salvus mesh)')
        moho_idx = int(moho_idx)
    if layers == "crust":
        return [int(x) for x in mesh_layers[:moho_idx]]
    if fluid is None:
        raise ValueError(f'layers="{layers}" needs the mesh\'s `fluid` elemental field')
    fluid_elements = np.where(np.asarray(fluid) == 1)[0]
    if len(fluid_elements) == 0:
        raise ValueError(f'layers="{layers}": the mesh has no fluid element (no outer core)')
    o_core_layer = np.asarray(layer_ids)[fluid_elements[0]]
    o_core_idx = int(np.where(mesh_layers == int(o_core_layer))[0][0])
    if layers == "mantle":
        picked = mesh_layers[moho_idx:o_core_idx]
    elif layers == "core":
        picked = mesh_layers[o_core_idx:]
    else:   # "nocore"
        picked = mesh_layers[:o_core_idx]
    return [int(x) for x in picked]


def _h5py_or_none():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def load_stored_layer_operator(stored_array):
    """``interp_info`` of the layered drivers: ``coeffs/<layer>`` and ``elements/<layer>`` datasets of
    ``interp_info.h5`` (reference interpolator.py:1035-1044) when h5py is importable -- a cache the reference wrote
    is read as it stands --, else (or when only that file exists) the same keys in ``interp_info.npz``."""
    if not stored_array:
        return None
    h5, npz = os.path.join(stored_array, "interp_info.h5"), os.path.join(stored_array, "interp_info.npz")
    h5py = _h5py_or_none()
    if h5py is not None and os.path.exists(h5):
        with h5py.File(h5, "r") as f:
            return ({k: f["elements"][k][:] for k in f["elements"].keys()},
                    {k: f["coeffs"][k][:] for k in f["coeffs"].keys()})
    if os.path.exists(npz):
        with np.load(npz) as f:
            return ({k.split("/", 1)[1]: f[k] for k in f.files if k.startswith("elements/")},
                    {k.split("/", 1)[1]: f[k] for k in f.files if k.startswith("coeffs/")})
    if os.path.exists(h5):
        raise ImportError(f"{h5} exists but h5py is not importable here: cannot read the stored operator")
    return None


def save_stored_layer_operator(stored_array, elements, coeffs):
    """Writes ``interp_info.h5`` in the reference's layout (interpolator.py:1061-1066) when h5py is importable,
    ``interp_info.npz`` with the same keys otherwise."""
    os.makedirs(stored_array, exist_ok=True)
    h5py = _h5py_or_none()
    if h5py is not None:
        with h5py.File(os.path.join(stored_array, "interp_info.h5"), "w") as f:
            for k in coeffs.keys():
                f.create_dataset(f"coeffs/{k}", data=coeffs[k])
            for k in elements.keys():
                f.create_dataset(f"elements/{k}", data=elements[k])
        return
    arrays = {f"elements/{k}": v for k, v in elements.items()}
    arrays.update({f"coeffs/{k}": v for k, v in coeffs.items()})
    np.savez(os.path.join(stored_array, "interp_info.npz"), **arrays)


def interpolate_gll_to_gll_layered(mesh_a: GllMesh, layer_a, target_gll_points, layer_b, params_to_interp,
                                   layers="all", nelem_to_search=30, tolerance=1.05, stored_array=None,
                                   existing=None, context=None, fluid_a=None, moho_idx=None, acceptance="tolerance"):
    """The array core of ``gll_2_gll_layered_multi_two`` (reference interpolator.py:980-1082): for every
    layer, the unique element-nodal points of the TARGET elements of that layer are located among the
    SOURCE elements of the same layer only (a tree over just their centroids, :1053), with
    ``snap_to_nearest=True`` (:1057), and the values are scattered back into the rows of those target
    elements (:1079-1081).  ``layer_a`` / ``layer_b``: the ``layer`` elemental field of the two meshes.

    Per layer everything runs on the device: ``mm_unique_points`` -> ``mm_interpolate_gll`` on the layer's
    sub-meshes -> ``mm_scatter_elements``.  ``stored_array``: the per-layer operator is kept as
    ``interp_info.npz`` (``coeffs/<layer>``, ``elements/<layer>``) and re-applied when it exists.
    Returns f64[C, E_t, P_t]; rows of target elements outside ``layers`` keep ``existing`` (zeros when not
    given), like the fields of the reference's ``new_mesh``.

    ``layers`` may be an Earth preset ("crust", "mantle", "core", "nocore"): resolved on the SOURCE mesh as the
    reference does (``create_layer_mask(mesh=original_mesh, ...)``, :1019), from ``fluid_a`` (its ``fluid``
    elemental field) and ``moho_idx`` (its global string) -- see :func:`assess_layers`.
    ``acceptance="bbox"``: the acceptance loop of the two older drivers (``gll_2_gll_layered`` :288-439 and
    ``gll_2_gll_layered_multi`` :442-618, through ``fill_value_array`` / ``_check_if_inside_element`` with
    ``ignore_hard_elements=True``: bounding-box pre-test, |xi| <= 1.04, nearest-centre fallback) instead of
    ``get_element_weights(snap_to_nearest=True)``: ``mm_locate_gll_bbox`` + ``mm_gather_elem`` per layer."""
    if acceptance not in ("tolerance", "bbox"):
        raise ValueError("acceptance must be 'tolerance' or 'bbox'")
    ctx = context or default_context()
    tgt = np.ascontiguousarray(target_gll_points, dtype=np.float64)
    layer_a, layer_b = np.asarray(layer_a), np.asarray(layer_b)
    if layer_a.shape != (mesh_a.nelem,) or layer_b.shape != (tgt.shape[0],):
        raise ValueError("layer_a / layer_b must hold one layer number per element")
    params = list(params_to_interp)
    n_t, p_t, dim = tgt.shape
    out = ctx.zeros((len(params), n_t, p_t), np.float64) if existing is None else \
        ctx.to_device(np.ascontiguousarray(existing, dtype=np.float64))
    if out.shape != (len(params), n_t, p_t):
        raise ValueError("existing must be [C, E_t, P_t]")
    stored = load_stored_layer_operator(stored_array)
    if stored is not None:
        print("No need for looping, we have the matrices")
    elements, coeffs = {}, {}
    for layer in assess_layers(layer_a, layers, fluid=fluid_a, moho_idx=moho_idx):
        key = str(layer)
        src_mask, tgt_mask = layer_a == layer, layer_b == layer
        if not tgt_mask.any():
            continue
        if not src_mask.any():
            raise ValueError(f"layer {layer} has target elements but no source elements")
        src = np.ascontiguousarray(mesh_a.gll_points[src_mask])
        fields = np.stack([mesh_a.element_nodal_fields[p][src_mask] for p in params])
        uniq, inv = ctx.unique_points(np.ascontiguousarray(tgt[tgt_mask]).reshape(-1, dim))
        if stored is not None:
            elements[key], coeffs[key] = stored[0][key], stored[1][key]
            vals = ctx.gather_elem(fields, elements[key], coeffs[key])
        elif acceptance == "bbox":
            print(f"Interpolating layer: {layer}")
            # (the older drivers: a tree over the layer's element centroids, nelem_to_search candidates, the
            # bounding-box loop; "hard" points -- final transform NaN -- keep the reference's constant xi)
            tree = ctx.knn_build(np.ascontiguousarray(src.mean(axis=1)))
            nn = tree.query(uniq, min(nelem_to_search, src.shape[0]))
            el, co, _hard = ctx.locate_gll_bbox(mesh_a.shape_order, nn, src, uniq)
            vals = ctx.gather_elem(fields, el, co)
            if stored_array:
                elements[key], coeffs[key] = el.numpy(), co.numpy()
        else:
            print("interpolating layer", layer, "...")
            if stored_array:
                vals, el, co, missing = ctx.interpolate_gll(mesh_a.shape_order, src, uniq, fields,
                                                            nelem_to_search=nelem_to_search, tolerance=tolerance,
                                                            snap_to_nearest=True, want_operator=True)
                elements[key], coeffs[key] = el.numpy(), co.numpy()
            else:
                vals, missing = ctx.interpolate_gll(mesh_a.shape_order, src, uniq, fields,
                                                    nelem_to_search=nelem_to_search, tolerance=tolerance,
                                                    snap_to_nearest=True)
            if missing:
                print(missing, "points of layer", layer, "could not find an enclosing element")
        ctx.scatter_elements(vals, inv, np.nonzero(tgt_mask)[0].astype(np.int64), out)
    if stored is None and stored_array:
        print("Saving interpolation matrices")
        save_stored_layer_operator(stored_array, elements, coeffs)
    return out.numpy()


def fix_fluid_solid(values, previous_values, solid_elements, parameters, context=None):
    """The fluid/solid fix-up at the end of ``gll_2_gll`` (reference interpolator.py:829-841) as a device
    pass: ``values`` / ``previous_values`` f64[E, nparam, P] (the ``MODEL/data`` layout), ``solid_elements``
    bool[E].  Fluid elements keep their previous values; so does a solid element whose VS (or VSV) came
    out exactly zero somewhere.  Returns the fixed array."""
    ctx = context or default_context()
    parameters = list(parameters)
    vs_index = parameters.index("VS") if "VS" in parameters else parameters.index("VSV")
    v = ctx.to_device(np.ascontiguousarray(values, dtype=np.float64))
    print("If any fluid values accidentally went to the solid part we fix it")
    ctx.fluid_solid_fix(v, np.ascontiguousarray(previous_values, dtype=np.float64), np.asarray(solid_elements, dtype=bool),
                        vs_index)
    return v.numpy()


def find_gll_centroids(gll_coordinates, dimensions=3):
    """The reference's ``_find_gll_centroids`` (interpolator.py:1389-1406): per-dimension
    ``np.mean(gll_coordinates[:, :, d], axis=1)`` -- NumPy's pairwise row sum over the strided view,
    not the node-order sum of ``mean(axis=1)`` on the 3-D array that the mesh reader uses; kept on the
    host in NumPy so that the tree is built over bit-identical centroids (an O(E P) pass)."""
    gll_coordinates = np.asarray(gll_coordinates, dtype=np.float64)
    if dimensions != gll_coordinates.shape[2]:
        raise ValueError("Dimensions of GLL model not the same as input")
    centroids = np.zeros(shape=[gll_coordinates.shape[0], dimensions])
    for d in range(dimensions):
        centroids[:, d] = np.mean(gll_coordinates[:, :, d], axis=1, dtype=np.float64)
    return centroids


def interpolate_gll_to_nodes(gll_points, gll_data, points, shape_order=4, nelem_to_search=20, context=None):
    """The array core of ``gll_2_exodus`` (reference interpolator.py:227-285): centroid tree over the
    GLL elements, ``nelem_to_search`` nearest per mesh node, the bounding-box acceptance loop
    ``_check_if_inside_element`` (:1409-1473) and ``np.sum(gll_data[element, :, :] * coeffs, axis=1)``.
    ``gll_points`` f64[E, P, dim], ``gll_data`` f64[E, C, P] (the layout of the HDF5 ``MODEL/data``),
    ``points`` f64[N, dim] -> values f64[N, C]."""
    ctx = context or default_context()
    gll_points = np.ascontiguousarray(gll_points, dtype=np.float64)
    dim = gll_points.shape[2]
    tree = ctx.knn_build(find_gll_centroids(gll_points, dim))
    pts = ctx.asdevice(np.ascontiguousarray(points, dtype=np.float64), np.float64)
    nn = tree.query(pts, nelem_to_search)
    elem, coeffs, _ = ctx.locate_gll_bbox(shape_order, nn, gll_points, pts)
    fields = np.ascontiguousarray(np.asarray(gll_data, dtype=np.float64).transpose(1, 0, 2))   # [C, E, P]
    return ctx.gather_elem(fields, elem, coeffs).numpy()


def _gll_operator_over_all_points(ctx, gll_points, points, nelem_to_search, ignore_hard_elements):
    """``find_gll_coeffs`` as ``query_model`` and ``gll_2_gll`` drive it (reference interpolator.py:91-126,
    :742-786): a tree over ALL GLL points (not the centroids), the ``nelem_to_search`` nearest points per
    coordinate mapped to their elements by ``floor(index / P)`` (an element can appear several times in a
    list), then the bounding-box acceptance loop (:1409-1473).  Returns device arrays ``(element, coeffs)``."""
    nelem, P, dim = gll_points.shape
    gll_order = int(round(P ** (1.0 / dim))) - 1
    tree = ctx.knn_build(gll_points.reshape(nelem * P, dim))
    pts = ctx.asdevice(points if hasattr(points, "numpy") else np.ascontiguousarray(points, dtype=np.float64),
                       np.float64)
    nearest = tree.query(pts, nelem_to_search)
    check(ctx.lib.mm_points_to_elements(ctx.handle, nearest.ptr, nearest.size, P), "mm_points_to_elements")   # floor(index / P), on the device
    elem, coeffs, hard = ctx.locate_gll_bbox(gll_order, nearest, gll_points, pts)
    if hard and not ignore_hard_elements:
        raise ValueError("Can't find an appropriate element.")
    return elem, coeffs


def query_gll_model(gll_points, gll_data, coordinates, nelem_to_search=20, ignore_hard_elements=False, context=None):
    """The array core of ``query_model`` (reference interpolator.py:60-139) after its file read and
    ``latlondepth_to_xyz``: :func:`_gll_operator_over_all_points`, then
    ``np.sum(original_data[elements] * coeffs, axis=2)``.
    ``gll_points`` f64[E, P, dim], ``gll_data`` f64[E, C, P], ``coordinates`` f64[N, dim] (Cartesian)
    -> values f64[N, C].  Like the reference it raises ``ValueError`` when no candidate element
    admits an inverse transform, unless ``ignore_hard_elements``.  Equidistant points (the copies of
    a node shared by several elements) are ordered by index here; cKDTree's order among them is
    unspecified."""
    ctx = context or default_context()
    gll_points = np.ascontiguousarray(gll_points, dtype=np.float64)
    elem, coeffs = _gll_operator_over_all_points(ctx, gll_points, coordinates, nelem_to_search, ignore_hard_elements)
    fields = np.ascontiguousarray(np.asarray(gll_data, dtype=np.float64).transpose(1, 0, 2))   # [C, E, P]
    return ctx.gather_elem(fields, elem, coeffs).numpy()


# ---------------------------------------------------------------------------------------------------
# File-level drivers: the reference's names and arguments; files through multimesh_amd.io (h5py for HDF5
# paths, scipy's netCDF reader for classic Exodus files), or already open h5py-like / mesh objects in
# place of the paths.  The work itself is the array cores above.
def query_model(coordinates, model, nelem_to_search=20, parameters="TTI", model_path="MODEL/data",
                coordinates_path="MODEL/coordinates", context=None):
    """Model parameters at ``coordinates`` f64[N, 3] = (latitude, longitude, depth in m) from a Salvus GLL
    model file (reference api.py:13-58, interpolator.py:60-139) -> f64[N, nparam] in the file's parameter
    order.  ``parameters`` is accepted and ignored, as in the reference."""
    from . import io as mio

    start = time.time()
    points, data, _ = mio.load_hdf5_params_to_memory(model, model_path, coordinates_path)
    coordinates = np.asarray(coordinates, dtype=np.float64)
    assert coordinates.ndim == 2 and coordinates.shape[1] == 3, "Make sure coordinates array has shape N,3"
    values = query_gll_model(points, data, latlondepth_to_xyz(coordinates), nelem_to_search, context=context)
    _report(start)
    return values


def exodus_2_gll(mesh, gll_model, gll_order=4, dimensions=3, nelem_to_search=20, parameters="TTI",
                 model_path="MODEL/data", coordinates_path="MODEL/coordinates", context=None):
    """Nodal parameters of an exodus hex8 mesh onto the GLL points of an HDF5 model, written to
    ``gll_model[model_path]`` as f64[nelem, nparam, P] with fresh dimension labels (reference api.py:61-103,
    interpolator.py:142-224; ``gll_order`` / ``dimensions`` are read off the coordinates).  ``mesh``: an
    Exodus file or a mesh object with ``points``, ``connectivity``, ``get_nodal_field``; ``gll_model``: an HDF5
    file or an open writable h5py-like object."""
    from . import io as mio

    start = time.time()
    exodus = mio.Exodus(mesh) if isinstance(mesh, (str, os.PathLike)) else mesh
    parameters = mio.pick_parameters(parameters)
    mesh_a = HexMesh(exodus.points, exodus.connectivity, {p: exodus.get_nodal_field(p) for p in parameters})
    with mio.open_h5(gll_model, "r+") as gll:
        gll_coords = np.array(gll[coordinates_path][:], dtype=np.float64)
        values, nfailed = interpolate_hex8_to_gll(mesh_a, gll_coords, parameters, nelem_to_search, context=context,
                                                  return_nfailed=True)
        assert nfailed == 0, f"{nfailed} points could not be interpolated."
        mio.remove_and_create_empty_dataset(gll, parameters, model_path, coordinates_path)
        gll[model_path][:, :, :] = values.transpose(1, 0, 2)
    _report(start)


def gll_2_exodus(gll_model, exodus_model, gll_order=4, dimensions=3, nelem_to_search=20, parameters="TTI",
                 model_path="MODEL/data", coordinates_path="MODEL/coordinates", gradient=False, context=None):
    """Every parameter of a GLL model onto the nodes of an exodus mesh, attached as its nodal fields
    (reference api.py:277-317, interpolator.py:227-285; like the reference, ``parameters`` is replaced by
    the model's own list and the exodus variables must exist).  ``exodus_model``: an Exodus file (opened
    in mode "a") or a mesh object with ``points`` and ``attach_field``."""
    from . import io as mio

    start = time.time()
    with mio.open_h5(gll_model, "r") as gll:
        gll_points = np.array(gll[coordinates_path][:], dtype=np.float64)
        gll_data = np.array(gll[model_path][:])
        parameters = mio.dimension_labels(gll[model_path], 1)
    exodus = mio.Exodus(exodus_model, mode="a") if isinstance(exodus_model, (str, os.PathLike)) else exodus_model
    shape_order = int(round(gll_points.shape[1] ** (1.0 / gll_points.shape[2]))) - 1
    values = interpolate_gll_to_nodes(gll_points, gll_data, exodus.points, shape_order, nelem_to_search, context)
    for i, param in enumerate(parameters):
        exodus.attach_field(param, values[:, i])
    _report(start)


def gll_2_gll(from_gll, to_gll, nelem_to_search=20, parameters="ISO", from_model_path="MODEL/data",
              to_model_path="MODEL/data", from_coordinates_path="MODEL/coordinates",
              to_coordinates_path="MODEL/coordinates", gradient=False, stored_array=None, context=None):
    """All parameters of one GLL model onto the GLL points of another, written to ``to_gll[to_model_path]``
    (reference api.py:106-155, interpolator.py:621-852): unique target points (device ``np.unique``),
    :func:`_gll_operator_over_all_points` with hard elements ignored, ``values[recon]`` scattered back, fluid
    elements and solid elements that caught a zero VS keep their previous values unless ``gradient``.  Like
    the reference, ``parameters`` is replaced by the source model's own list.  ``stored_array``: directory of
    ``elements.npy`` + ``coeffs.npy``; ``coeffs.npy`` is written as f64[1, P, U] -- the reference writes
    ``nparam`` identical copies [nparam, P, U] and broadcasts either on load -- and both are read."""
    from . import io as mio

    start = time.time()
    ctx = context or default_context()
    print("Initialization stage")
    print(f"Stored array: {stored_array}")
    original_points, original_data, parameters = mio.load_hdf5_params_to_memory(from_gll, from_model_path,
                                                                                from_coordinates_path)
    with mio.open_h5(to_gll, "r+") as new:
        new_points = np.array(new[to_coordinates_path][:], dtype=np.float64)
        elem_params = mio.dimension_labels(new["MODEL/element_data"], 1)
        fluid_elements = np.array(new["MODEL/element_data"][:, elem_params.index("fluid")]).astype(bool)
        solid_elements = np.invert(fluid_elements)
        new_values = np.array(new[to_model_path][:], dtype=np.float64)
        unique_new_points, recon = ctx.unique_points(new_points.reshape(-1, new_points.shape[2]))
        stored = load_stored_operator(stored_array)
        if stored is not None:
            print("Matrix was already stored. Will use that one")
            element, coeffs = stored
            element = np.asarray(element).astype(np.int64)
            if coeffs.ndim == 3:                       # [nparam or 1, P, U] as the reference stores it
                coeffs = np.ascontiguousarray(coeffs[0].T)
        else:
            print("Now we start interpolating")
            element, coeffs = _gll_operator_over_all_points(ctx, np.ascontiguousarray(original_points),
                                                            unique_new_points, nelem_to_search, True)
            if stored_array:
                save_stored_operator(stored_array, element.numpy(), coeffs.numpy().T[None, :, :])
        fields = np.ascontiguousarray(np.asarray(original_data, dtype=np.float64).transpose(1, 0, 2))   # [C, E, P]
        unique_values = ctx.gather_elem(fields, element, coeffs).numpy()                                # [U, C]
        values = np.ascontiguousarray(unique_values[recon.numpy()].reshape(new_points.shape[0], new_points.shape[1],
                                                                          len(parameters)).swapaxes(1, 2))
        if not gradient:
            if new_values.shape != values.shape:
                raise ValueError("the receiving model must already hold the source model's parameters "
                                 f"({values.shape[1]}), it has {new_values.shape[1]}")
            values = fix_fluid_solid(values, new_values, solid_elements, parameters, context=ctx)
        mio.remove_and_create_empty_dataset(new, parameters, to_model_path, to_coordinates_path)
        new[to_model_path][:, :, :] = values
    _report(start)


def gll_2_gll_layered_multi_two(from_gll, to_gll, layers, nelem_to_search=30, parameters="all", stored_array=None,
                                make_spherical=False, tolerance=1.05, context=None):
    """Layer by layer, GLL model to GLL model, through the fast Salvus-mesh reader (reference api.py:645-699,
    interpolator.py:980-1082): the ``layer`` elemental field of both meshes, :func:`interpolate_gll_to_gll_layered`,
    every parameter attached to ``to_gll``.  ``layers``: "all" or a list of layer numbers (the Earth presets
    need mesh metadata); ``make_spherical`` is Earth-specific and refused."""
    from . import io as mio

    if make_spherical:
        raise NotImplementedError("map_to_sphere (reference interpolator.py:1085-1144) is Earth-specific "
                                  "(out of scope, SURVEY.md §8)")
    start = time.time()
    original_mesh = mio.SalvusMesh(from_gll, fast_mode=False)
    new_mesh = mio.SalvusMesh(to_gll, fast_mode=False)
    if isinstance(parameters, str) and parameters == "all":
        parameters = list(original_mesh.element_nodal_fields.keys())
    parameters = mio.pick_parameters(parameters)
    mesh_a = GllMesh(original_mesh.points, original_mesh.shape_order,
                     {p: original_mesh.element_nodal_fields[p] for p in parameters})
    existing = np.stack([new_mesh.element_nodal_fields[p] for p in parameters])
    values = interpolate_gll_to_gll_layered(mesh_a, original_mesh.elemental_fields["layer"], new_mesh.points,
                                            new_mesh.elemental_fields["layer"], parameters, layers=layers,
                                            nelem_to_search=nelem_to_search, tolerance=tolerance,
                                            stored_array=stored_array, existing=existing, context=context,
                                            **_layer_metadata(original_mesh))
    for i, param in enumerate(parameters):
        new_mesh.attach_field(name=param, data=values[i])
    _report(start)


def _layer_metadata(mesh):
    """What the Earth presets of :func:`assess_layers` read off a Salvus mesh (reference utils.py:413-429)."""
    moho = getattr(mesh, "global_strings", {}).get("moho_idx")
    if isinstance(moho, bytes):
        moho = moho.decode()
    return {"fluid_a": mesh.elemental_fields.get("fluid"), "moho_idx": None if moho is None else int(moho)}


def _gll_2_gll_layered_bbox(from_gll, to_gll, layers, nelem_to_search, parameters, stored_array, make_spherical,
                            keep_existing, context):
    from . import io as mio

    if make_spherical:
        raise NotImplementedError("map_to_sphere (reference interpolator.py:1085-1144) is Earth-specific "
                                  "(out of scope, SURVEY.md §8)")
    print("Initialization stage")
    original_mesh = mio.SalvusMesh(from_gll, fast_mode=False)
    new_mesh = mio.SalvusMesh(to_gll, fast_mode=False)
    if isinstance(parameters, str) and parameters == "all":
        parameters = list(original_mesh.element_nodal_fields.keys())
    parameters = mio.pick_parameters(parameters)
    mesh_a = GllMesh(original_mesh.points, original_mesh.shape_order,
                     {p: original_mesh.element_nodal_fields[p] for p in parameters})
    existing = np.stack([new_mesh.element_nodal_fields[p] for p in parameters]) if keep_existing else None
    values = interpolate_gll_to_gll_layered(mesh_a, original_mesh.elemental_fields["layer"], new_mesh.points,
                                            new_mesh.elemental_fields["layer"], parameters, layers=layers,
                                            nelem_to_search=nelem_to_search, stored_array=stored_array,
                                            existing=existing, context=context, acceptance="bbox",
                                            **_layer_metadata(original_mesh))
    for i, param in enumerate(parameters):
        new_mesh.attach_field(name=param, data=values[i])


def gll_2_gll_layered(from_gll, to_gll, layers, nelem_to_search=20, parameters="ISO", stored_array=None,
                      make_spherical=False, context=None):
    """Layer by layer with the bounding-box acceptance loop (reference api.py:158-211, interpolator.py:288-439);
    elements of ``to_gll`` outside ``layers`` come out ZERO, as the reference's ``np.zeros_like`` fields do
    (:421).  Superseded in the reference by :func:`gll_2_gll_layered_multi_two`."""
    start = time.time()
    _gll_2_gll_layered_bbox(from_gll, to_gll, layers, nelem_to_search, parameters, stored_array, make_spherical,
                            keep_existing=False, context=context)
    _report(start)


def gll_2_gll_layered_multi(from_gll, to_gll, layers="nocore", nelem_to_search=20, parameters="all", threads=None,
                            stored_array=None, make_spherical=False, context=None):
    """The same per layer in parallel (reference api.py:214-274, interpolator.py:442-618: a process pool over
    the layers -- here every layer is a device pass, ``threads`` is accepted and ignored); elements outside
    ``layers`` keep the values ``to_gll`` holds (:606)."""
    start = time.time()
    _gll_2_gll_layered_bbox(from_gll, to_gll, layers, nelem_to_search, parameters, stored_array, make_spherical,
                            keep_existing=True, context=context)
    _report(start)
