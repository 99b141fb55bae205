import numpy as np

from multimesh_amd import synth


def test_hex_mesh_layout():
    n = 5
    p, c = synth.hex_mesh(n, seed=1)
    assert p.shape == (n ** 3, 3) and c.shape == ((n - 1) ** 3, 8)
    assert c.dtype == np.int64 and p.flags.c_contiguous and c.flags.c_contiguous
    # boundary nodes stay on the unit-cube faces; hulls of different seeds coincide
    grid = p.reshape(n, n, n, 3)
    assert np.array_equal(grid[0, :, :, 0], np.zeros((n, n)))
    assert np.array_equal(grid[-1, :, :, 0], np.ones((n, n)))
    # exodus order: bottom face counter-clockwise seen from +z, positive volume
    v = p[c[0]]
    vol = np.dot(np.cross(v[1] - v[0], v[3] - v[0]), v[4] - v[0])
    assert vol > 0
    r = synth.reorder_hex8(c)
    assert np.array_equal(r[:, 1], c[:, 3]) and np.array_equal(r[:, 3], c[:, 1])
    assert np.array_equal(r[:, [0, 2, 4, 5, 6, 7]], c[:, [0, 2, 4, 5, 6, 7]])


def test_config_sizes_match_survey():
    # SURVEY.md §8 header: cfg2 n=101 -> 1,030,301 nodes / 1,000,000 elements; cfg3 n=216
    assert 101 ** 3 == 1030301 and 100 ** 3 == 1000000
    assert 216 ** 3 == 10077696 and 215 ** 3 == 9938375
    assert synth.CONFIGS["metric"]["n_src"] == 216 and synth.CONFIGS["metric"]["ncomp"] == 1


def test_hex_mesh_rows_equal_slices_of_the_full_mesh():
    # cfg4 shards (bench.py --workload cfg4, tests/test_parity_gpu.py) are generated without the 2.4 GB mesh
    from multimesh_amd.distributed import shard_bounds

    for n, seed in ((2, 3), (3, 7), (9, 7), (12, 11)):
        full, _ = synth.hex_mesh(n, seed=seed)
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, b = shard_bounds(n ** 3, world, r)
                assert np.array_equal(synth.hex_mesh_rows(n, a, b, seed=seed), full[a:b])
        for a, b in ((0, 0), (5, 6), (n ** 3 - 1, n ** 3), (n * n - 1, n * n + 1)):
            if b <= n ** 3:
                assert np.array_equal(synth.hex_mesh_rows(n, a, b, seed=seed), full[a:b])
    assert synth.CONFIGS["cfg4"]["n_tgt"] ** 3 == 100_544_625
