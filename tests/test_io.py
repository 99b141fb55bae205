"""Mesh file adapters: label parsing, the HDF5 layout logic on an in-memory h5py stand-in
(h5py is not in this image), and classic-netCDF Exodus files written and read back through scipy."""
import numpy as np
import pytest

from multimesh_amd import io as mio, synth


def test_dimension_labels_as_salvus_writes_them():
    assert mio.parse_dimension_labels(b"[ VP | VS | RHO ]") == ["VP", "VS", "RHO"]
    assert mio.parse_dimension_labels("[ fluid | layer ]") == ["fluid", "layer"]
    assert mio.parse_dimension_labels(np.bytes_(b"[ VPgrad | VSgrad ]"), drop_grad=True) == ["VP", "VS"]
    assert mio.parse_dimension_labels("[ QMU ]") == ["QMU"]
    assert mio.pick_parameters("ISO") == ["QKAPPA", "QMU", "RHO", "VP", "VS"]
    assert mio.pick_parameters("TTI")[:4] == ["VPV", "VPH", "VSV", "VSH"] and mio.pick_parameters(["A"]) == ["A"]


def _model(nelem=6, order=2, params=("VP", "VS")):
    P = (order + 1) ** 3
    rng = np.random.default_rng(3)
    h = mio.MemoryH5()
    h.create_dataset("MODEL/coordinates", data=rng.uniform(size=(nelem, P, 3)))
    ds = h.create_dataset("MODEL/data", data=rng.normal(size=(nelem, len(params), P)))
    mio.set_dimension_labels(ds, list(params))
    ed = h.create_dataset("MODEL/element_data", data=np.stack([np.arange(nelem) % 2, np.arange(nelem) // 2], axis=1) * 1.0)
    ed.attrs["DIMENSION_LABELS"] = np.array([b"element", b"[ fluid | layer ]"])
    h["MODEL"].attrs["reference_frame"] = np.bytes_(b"cartesian")
    return h


def test_salvus_mesh_reader_on_the_hdf5_layout():
    h = _model()
    m = mio.SalvusMesh(h, fast_mode=False)
    assert (m.nelem, m.n_gll_points, m.dimensions, m.shape_order) == (6, 27, 3, 2)
    assert m.nodal_parameter_indices == ["VP", "VS"] and m.elemental_parameter_indices == ["fluid", "layer"]
    assert np.array_equal(m.element_nodal_fields["VS"], h["MODEL/data"][:, 1, :])
    assert np.array_equal(m.elemental_fields["layer"], [0, 0, 1, 1, 2, 2])
    assert np.array_equal(m.get_element_centroids(), m.points.mean(axis=1))
    assert np.array_equal(m.get_element_nodal_field("VP"), h["MODEL/data"][:, 0, :])
    assert m.global_strings == {"reference_frame": b"cartesian"}
    new = np.full((6, 27), 7.0)
    m.attach_field("VS", new)
    assert np.array_equal(h["MODEL/data"][:, 1, :], new) and np.array_equal(m.element_nodal_fields["VS"], new)
    m.attach_field("layer", np.arange(6.0))
    assert np.array_equal(h["MODEL/element_data"][:, 1], np.arange(6.0))
    with pytest.raises(ValueError):
        m.attach_field("NEW", new)
    with pytest.raises(ValueError):
        m.attach_field("VS", np.zeros((6, 5)))


def test_load_params_and_fresh_dataset():
    h = _model(params=("VPgrad", "RHOgrad", "VSgrad"))
    points, data, params = mio.load_hdf5_params_to_memory(h)
    assert params == ["VP", "RHO", "VS"] and data.shape == (6, 3, 27) and points.dtype == np.float64
    ds = mio.remove_and_create_empty_dataset(h, ["A", "B"])
    assert h["MODEL/data"].shape == (6, 2, 27) and not np.any(ds[()]) and mio.dimension_labels(h["MODEL/data"]) == ["A", "B"]


def test_paths_need_h5py_and_say_so(tmp_path):
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            mio.SalvusMesh(str(tmp_path / "mesh.h5"))


def test_exodus_classic_file_round_trip(tmp_path):
    p, c = synth.hex_mesh(4, seed=1)
    fn = str(tmp_path / "mesh.e")
    mio.write_exodus_classic(fn, p, c, {"VP": 2 * p[:, 0], "RHO": p[:, 1]}, {"layer": np.arange(len(c), dtype=float)})
    from scipy.io import netcdf_file
    with netcdf_file(fn, "r", mmap=False) as f:          # the file's own convention: 1-based connectivity
        assert f.variables["connect1"][:].min() == 1 and f.variables["connect1"][:].max() == len(p)
    e = mio.Exodus(fn, mode="a")
    assert (e.ndim, e.nelem, e.nodes_per_element, e.npoint) == (3, 27, 8, 64)
    assert e.connectivity.dtype == np.int64 and np.array_equal(e.connectivity, c) and np.array_equal(e.points, p)
    assert e.nodal_parameters == ["VP", "RHO"] and e.elem_var_names == ["layer"]
    assert np.array_equal(e.get_nodal_field("RHO"), p[:, 1])
    e.attach_field("VP", p[:, 2])
    e.attach_field("layer", np.ones(len(c)))
    again = mio.Exodus(fn)
    assert np.array_equal(again.get_nodal_field("VP"), p[:, 2]) and np.array_equal(again.get_nodal_field("RHO"), p[:, 1])
    assert np.array_equal(again.get_element_field("layer"), np.ones(len(c)))
    with pytest.raises(ValueError):
        e.attach_field("UNKNOWN", p[:, 0])
    with pytest.raises(AssertionError):
        again.attach_field("VP", p[:, 0])                # opened read-only


def test_exodus_2d_quads(tmp_path):
    n = 5
    xy = np.stack(np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n), indexing="ij"), axis=-1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    conn = np.stack([a, a + n, a + n + 1, a + 1], axis=1)
    fn = str(tmp_path / "quads.e")
    mio.write_exodus_classic(fn, xy, conn, {"T": xy[:, 0]})
    e = mio.Exodus(fn)
    assert e.ndim == 2 and e.nodes_per_element == 4 and np.array_equal(e.connectivity, conn) and np.array_equal(e.points, xy)


def test_memory_dataset_takes_masks_and_fancy_indices_like_h5py():
    from multimesh_amd import io as mio

    h = mio.MemoryH5()
    d = h.create_dataset("MODEL/data", data=np.arange(24.0).reshape(4, 3, 2))
    mask = np.array([True, False, True, False])
    assert np.array_equal(d[mask], np.arange(24.0).reshape(4, 3, 2)[mask])
    assert np.array_equal(d[[0, 3]], np.arange(24.0).reshape(4, 3, 2)[[0, 3]])
    assert np.array_equal(d[()], np.arange(24.0).reshape(4, 3, 2)) and d[()] is not d[()]
