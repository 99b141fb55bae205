"""Targets sharded over the GPUs of one node, source mesh replicated.

Each target's result depends only on the read-only source mesh -- the reference itself already
splits targets into independent chunks across worker processes
(``components/interpolator.py:1239-1251``) -- so the data path has NO collective until the end:
rank r interpolates the contiguous range ``[r*ceil(N/G), (r+1)*ceil(N/G))`` and ONE all-gather
reassembles the field (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPUs for tests).  The
failed-point count is all-reduced.  Because the per-target computation does not depend on the
partition, the gathered field is bit-identical for every world size.

One process per GPU (``torch.distributed``); torch is plumbing here (device memory + process
group), the computation is the C-ABI library.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total: int, world_size: int, rank: int):
    """Contiguous range of targets owned by ``rank``: chunks of ceil(N/G), last ones short/empty."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    chunk = -(-n_total // world_size) if n_total > 0 else 0
    start = min(rank * chunk, n_total)
    stop = min(start + chunk, n_total)
    return start, stop


def all_gather_rows(local, n_total: int, group=None):
    """All-gather row blocks of unequal length (shard_bounds layout) into the full [n_total, ...]
    tensor: blocks are padded to the common chunk so that ONE all_gather_into_tensor suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    chunk = -(-n_total // world) if n_total > 0 else 0
    tail = tuple(local.shape[1:])
    if local.shape[0] > chunk:
        raise ValueError("local block longer than ceil(N/world)")
    if chunk == 0:
        return local.new_empty((0,) + tail)
    if local.shape[0] == chunk and local.is_contiguous():
        padded = local
    else:
        padded = local.new_zeros((chunk,) + tail)
        padded[: local.shape[0]] = local
    out = local.new_empty((world * chunk,) + tail)
    dist.all_gather_into_tensor(out, padded, group=group)
    return out[:n_total]


def interpolate_sharded(points, local_interpolate, group=None):
    """Shard ``points`` (f64[N, 3], identical on every rank) by :func:`shard_bounds`, run
    ``local_interpolate(points_shard) -> (values[n_local, C] tensor, nfailed)`` on this rank's
    range, all-gather the values and all-reduce the failure count.

    Returns ``(values[N, C] tensor, nfailed_total)`` on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = int(points.shape[0])
    start, stop = shard_bounds(n_total, world, rank)
    values, nfailed = local_interpolate(points[start:stop])
    if values.shape[0] != stop - start:
        raise ValueError("local_interpolate returned the wrong number of rows")
    gathered = all_gather_rows(values, n_total, group)
    nf = torch.tensor([int(nfailed)], dtype=torch.int64, device=values.device)
    dist.all_reduce(nf, group=group)
    return gathered, int(nf.item())


class HipShardInterpolator:
    """``local_interpolate`` for :func:`interpolate_sharded` backed by the HIP library: keeps the
    replicated source mesh resident on this rank's GPU and runs the fused hot path per shard."""

    def __init__(self, nodes, connectivity, fields, nelem_to_search=20, device_index=None):
        import torch

        from .device import Context

        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        self.ctx = Context(device_index, stream=torch.cuda.current_stream(self.device).cuda_stream)
        self.k = nelem_to_search
        as_dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(self.device)  # noqa: E731
        self.nodes = as_dev(nodes, np.float64)
        self.conn = as_dev(connectivity, np.int64)
        self.fields = as_dev(np.atleast_2d(fields), np.float64)

    def __call__(self, points_shard):
        import torch

        pts = torch.as_tensor(np.ascontiguousarray(points_shard, dtype=np.float64)).to(self.device) \
            if not hasattr(points_shard, "data_ptr") else points_shard.contiguous()
        out = torch.empty((pts.shape[0], self.fields.shape[0]), dtype=torch.float64, device=self.device)
        if pts.shape[0] == 0:
            return out, 0
        _, nfailed = self.ctx.interpolate_hex8(self.nodes, self.conn, pts, self.fields,
                                               nelem_to_search=self.k, out=out)
        return out, nfailed


class HipShardGllInterpolator:
    """The GLL counterpart (reference interpolator.py:931-977, whose per-process chunks of targets
    are this sharding, :1239-1251): element-nodal source mesh and fields replicated on this rank's
    GPU, the fused GLL path (``mm_interpolate_gll``) per shard of target points."""

    def __init__(self, gll_points, shape_order, element_nodal_fields, nelem_to_search=20, tolerance=1.05,
                 device_index=None):
        import torch

        from .device import Context

        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        self.ctx = Context(device_index, stream=torch.cuda.current_stream(self.device).cuda_stream)
        self.order, self.k, self.tolerance = int(shape_order), nelem_to_search, float(tolerance)
        as_dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)  # noqa: E731
        self.gll_points = as_dev(gll_points)                       # [E, P, dim]
        f = np.asarray(element_nodal_fields, dtype=np.float64)
        self.fields = as_dev(f[None] if f.ndim == 2 else f)        # [C, E, P]

    def __call__(self, points_shard):
        import torch

        pts = torch.as_tensor(np.ascontiguousarray(points_shard, dtype=np.float64)).to(self.device) \
            if not hasattr(points_shard, "data_ptr") else points_shard.contiguous()
        out = torch.empty((pts.shape[0], self.fields.shape[0]), dtype=torch.float64, device=self.device)
        if pts.shape[0] == 0:
            return out, 0
        _, missing = self.ctx.interpolate_gll(self.order, self.gll_points, pts, self.fields,
                                              nelem_to_search=self.k, tolerance=self.tolerance, out=out)
        return out, missing
