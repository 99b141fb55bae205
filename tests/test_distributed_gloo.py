"""world_size > 1 on CPU (gloo): the sharding / padding / all-gather / all-reduce layer of
multimesh_amd.distributed.  The per-shard computation is injected; here it is the CPU oracle
(tests may use it), on the GPU it is the HIP library (HipShardInterpolator).  The gathered field
must be bit-identical to the single-process result for every world size."""
import os
import socket

import numpy as np
import pytest

from multimesh_amd import synth
from multimesh_amd.distributed import shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 2, 7, 8, 9, 1001, 100_544_625):
        for g in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            chunk = -(-n // g) if n else 0
            assert all(e - s <= chunk for s, e in spans)
    # cfg4 = 100,544,625 targets -> 12,568,079 per GPU at 8 GPUs (ceil)
    assert shard_bounds(100_544_625, 8, 0) == (0, 12_568_079)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_targets, tmpdir):
    import torch
    import torch.distributed as dist

    from multimesh_amd.distributed import interpolate_sharded
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pa, ca = synth.hex_mesh(7, seed=1)
        rng = np.random.default_rng(5)
        pts = rng.uniform(-0.05, 1.05, size=(n_targets, 3))      # a few targets fall outside -> nfailed > 0
        fields = synth.vector_field(pa)
        cen = O.centroid(ca, pa)
        conn = synth.reorder_hex8(ca)

        def local(shard):
            if len(shard) == 0:
                return torch.zeros((0, 3), dtype=torch.float64), 0
            nn, _ = O.knn_ckdtree(cen, shard, 20)
            enc, w, nf = O.locate_hex8(nn, conn, pa, shard)
            return torch.from_numpy(O.gather(fields, enc, w)), nf

        vals, nfailed = interpolate_sharded(pts, local)
        np.save(os.path.join(tmpdir, f"vals_{world}_{rank}.npy"), vals.numpy())
        np.save(os.path.join(tmpdir, f"nf_{world}_{rank}.npy"), np.array([nfailed]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_targets", [(2, 1001), (3, 10), (2, 1), (4, 0)])
def test_sharded_result_is_partition_independent(tmp_path, world, n_targets):
    import torch.multiprocessing as mp

    from oracle import oracle as O

    mp.spawn(_worker, args=(world, _free_port(), n_targets, str(tmp_path)), nprocs=world, join=True)
    # single-process truth
    pa, ca = synth.hex_mesh(7, seed=1)
    pts = np.random.default_rng(5).uniform(-0.05, 1.05, size=(n_targets, 3))
    if n_targets:
        nn, _ = O.knn_ckdtree(O.centroid(ca, pa), pts, 20)
        enc, w, nf = O.locate_hex8(nn, synth.reorder_hex8(ca), pa, pts)
        truth = O.gather(synth.vector_field(pa), enc, w)
    else:
        truth, nf = np.zeros((0, 3)), 0
    for rank in range(world):
        got = np.load(tmp_path / f"vals_{world}_{rank}.npy")
        assert got.shape == truth.shape
        assert np.array_equal(got, truth)                        # bit-identical on every rank
        assert int(np.load(tmp_path / f"nf_{world}_{rank}.npy")[0]) == nf
