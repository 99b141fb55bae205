"""Synthetic meshes and fields for parity tests and the benchmark.

Unit-cube structured hexahedral grids with ``n`` nodes per side; node id =
``(i*n + j)*n + k`` with (i, j, k) along (x, y, z); element id =
``(i*(n-1) + j)*(n-1) + k``.  Interior nodes are jittered by ``U(-jitter, jitter)*h``
per axis (``numpy.random.default_rng(seed)``), boundary nodes are left in place so the
hulls of two meshes with different seeds coincide.

Element node order is the exodus hex8 order (counter-clockwise bottom face, then top
face) that the reference reads from its mesh files; ``reorder_hex8`` is the host-side
column permutation the reference applies before calling the locator
(reference scripts/cli.py:79-81).
"""
from __future__ import annotations

import numpy as np

# exodus hex8 corner offsets (di, dj, dk), nodes 0..7
_EXODUS_CORNERS = np.array(
    [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]],
    dtype=np.int64,
)


def reorder_hex8(connectivity: np.ndarray) -> np.ndarray:
    """exodus hex8 order -> the locator's corner order (reference scripts/cli.py:79-81)."""
    permutation = [0, 3, 2, 1, 4, 5, 6, 7]
    return np.ascontiguousarray(connectivity[:, np.argsort(permutation)])


def hex_mesh(n: int, seed: int = 1, jitter: float = 0.2, lo=(0.0, 0.0, 0.0),
             hi=(1.0, 1.0, 1.0)):
    """Returns (points f64[n^3, 3], connectivity int64[(n-1)^3, 8] in exodus order)."""
    if n < 2:
        raise ValueError("need at least 2 nodes per side")
    rng = np.random.default_rng(seed)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    h = (hi - lo) / (n - 1)
    ax = [lo[a] + h[a] * np.arange(n, dtype=np.float64) for a in range(3)]
    pts = np.empty((n, n, n, 3), dtype=np.float64)
    pts[..., 0] = ax[0][:, None, None]
    pts[..., 1] = ax[1][None, :, None]
    pts[..., 2] = ax[2][None, None, :]
    if n > 2 and jitter > 0:
        jit = rng.uniform(-jitter, jitter, size=(n - 2, n - 2, n - 2, 3)) * h
        pts[1:-1, 1:-1, 1:-1, :] += jit
    pts = pts.reshape(-1, 3)

    m = n - 1
    i, j, k = np.meshgrid(np.arange(m, dtype=np.int64), np.arange(m, dtype=np.int64),
                          np.arange(m, dtype=np.int64), indexing="ij")
    base = ((i * n + j) * n + k).reshape(-1)
    off = (_EXODUS_CORNERS[:, 0] * n + _EXODUS_CORNERS[:, 1]) * n + _EXODUS_CORNERS[:, 2]
    conn = base[:, None] + off[None, :]
    return np.ascontiguousarray(pts), np.ascontiguousarray(conn)


def hex_mesh_rows(n: int, start: int, stop: int, seed: int = 1, jitter: float = 0.2):
    """``hex_mesh(n, seed, jitter)[0][start:stop]`` without building the whole mesh: the rows of
    a contiguous shard of a big target mesh (cfg4: one rank's 12.6 M of the 465^3 = 100.5 M nodes).

    Only the x-planes the range touches are generated; the jitter stream is positioned with
    ``bit_generator.advance`` (``Generator.uniform`` draws one 64-bit word per double), so the
    result is bit-identical to slicing the full mesh (tests/test_synth.py)."""
    if n < 2 or not 0 <= start <= stop <= n ** 3:
        raise ValueError("need n >= 2 and 0 <= start <= stop <= n^3")
    if start == stop:
        return np.empty((0, 3), dtype=np.float64)
    plane = n * n
    i0, i1 = start // plane, (stop - 1) // plane + 1          # x-planes [i0, i1)
    h = 1.0 / (n - 1)
    ax = h * np.arange(n, dtype=np.float64)
    pts = np.empty((i1 - i0, n, n, 3), dtype=np.float64)
    pts[..., 0] = ax[i0:i1, None, None]
    pts[..., 1] = ax[None, :, None]
    pts[..., 2] = ax[None, None, :]
    if n > 2 and jitter > 0:
        a, b = max(i0, 1), min(i1, n - 1)                      # interior planes among them
        if b > a:
            rng = np.random.default_rng(seed)
            per_plane = (n - 2) * (n - 2) * 3
            rng.bit_generator.advance((a - 1) * per_plane)
            jit = rng.uniform(-jitter, jitter, size=(b - a, n - 2, n - 2, 3)) * h
            pts[a - i0:b - i0, 1:-1, 1:-1, :] += jit
    pts = pts.reshape(-1, 3)
    return np.ascontiguousarray(pts[start - i0 * plane:stop - i0 * plane])


def quad_mesh(n: int, seed: int = 1, jitter: float = 0.2):
    """2-D analogue: (points f64[n^2, 2], connectivity int64[(n-1)^2, 4]) counter-clockwise."""
    rng = np.random.default_rng(seed)
    h = 1.0 / (n - 1)
    ax = h * np.arange(n, dtype=np.float64)
    pts = np.empty((n, n, 2))
    pts[..., 0] = ax[:, None]
    pts[..., 1] = ax[None, :]
    if n > 2 and jitter > 0:
        pts[1:-1, 1:-1, :] += rng.uniform(-jitter, jitter, size=(n - 2, n - 2, 2)) * h
    pts = pts.reshape(-1, 2)
    m = n - 1
    i, j = np.meshgrid(np.arange(m, dtype=np.int64), np.arange(m, dtype=np.int64), indexing="ij")
    base = (i * n + j).reshape(-1)
    off = np.array([0, n, n + 1, 1], dtype=np.int64)  # (0,0) (1,0) (1,1) (0,1)
    return np.ascontiguousarray(pts), np.ascontiguousarray(base[:, None] + off[None, :])


def gll_nodes_1d(order: int) -> np.ndarray:
    """GLL nodes on [-1, 1] for the orders the reference supports (1, 2, 4)."""
    if order == 1:
        return np.array([-1.0, 1.0])
    if order == 2:
        return np.array([-1.0, 0.0, 1.0])
    if order == 4:
        a = np.sqrt(3.0 / 7.0)
        return np.array([-1.0, -a, 0.0, a, 1.0])
    raise ValueError("order must be 1, 2 or 4")


def gll_mesh(n: int, order: int, seed: int = 1, jitter: float = 0.2, dim: int = 3):
    """Element-nodal GLL mesh ``f64[nelem, (order+1)^dim, dim]`` (the layout of the reference's
    ``MODEL/coordinates`` / ``mesh.points[mesh.connectivity]``): the control nodes of every element
    of :func:`hex_mesh` / :func:`quad_mesh` placed by the (bi/tri)linear map of the tensor GLL
    points, node index p = i + (order+1) j + (order+1)^2 k with the first reference axis fastest."""
    g = gll_nodes_1d(order)
    m = order + 1
    if dim == 3:
        pts, conn = hex_mesh(n, seed=seed, jitter=jitter)
        v = pts[conn]                                  # exodus order corners [E, 8, 3]
        corner = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1],
                           [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=np.float64)
        k_, j_, i_ = np.meshgrid(g, g, g, indexing="ij")  # p = i + m j + m^2 k
        xi = np.stack([i_.ravel(), j_.ravel(), k_.ravel()], axis=1)            # [P, 3]
        shape = np.prod(1.0 + xi[:, None, :] * corner[None, :, :], axis=2) / 8.0  # [P, 8]
    elif dim == 2:
        pts, conn = quad_mesh(n, seed=seed, jitter=jitter)
        v = pts[conn]                                  # counter-clockwise corners [E, 4, 2]
        corner = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64)
        j_, i_ = np.meshgrid(g, g, indexing="ij")
        xi = np.stack([i_.ravel(), j_.ravel()], axis=1)
        shape = np.prod(1.0 + xi[:, None, :] * corner[None, :, :], axis=2) / 4.0
    else:
        raise ValueError("dim must be 2 or 3")
    assert shape.shape[0] == m ** dim
    return np.ascontiguousarray(np.einsum("pc,ecd->epd", shape, v))


def field_linear(p: np.ndarray) -> np.ndarray:
    """f1 = 1 + 2x - 3y + 0.5z: reproduced exactly (to the Newton tolerance) by hex8."""
    z = p[..., 2] if p.shape[-1] > 2 else 0.0
    return 1.0 + 2.0 * p[..., 0] - 3.0 * p[..., 1] + 0.5 * z


def field_smooth(p: np.ndarray) -> np.ndarray:
    """f2 = sin(2 pi x) cos(3 pi y) + z^2 (parity only)."""
    z = p[:, 2] if p.shape[1] > 2 else 0.0
    return np.sin(2 * np.pi * p[:, 0]) * np.cos(3 * np.pi * p[:, 1]) + z * z


def field_xyz(p: np.ndarray) -> np.ndarray:
    z = p[:, 2] if p.shape[1] > 2 else 1.0
    return p[:, 0] * p[:, 1] * z


def vector_field(p: np.ndarray) -> np.ndarray:
    """cfg3's 3-component field, component-major f64[3, M]."""
    return np.ascontiguousarray(np.stack([field_linear(p), field_smooth(p), field_xyz(p)]))


#: benchmark configurations of BASELINE.json / SURVEY.md §8 (nodes per side)
CONFIGS = {
    "cfg2": dict(n_src=101, n_tgt=101, ncomp=1),   # 1M -> 1M
    "cfg3": dict(n_src=216, n_tgt=216, ncomp=3),   # 10M -> 10M, vector field
    "metric": dict(n_src=216, n_tgt=216, ncomp=1),  # BASELINE.json metric: 10M -> 10M, 1 scalar
    "cfg4": dict(n_src=216, n_tgt=465, ncomp=1),   # 100M targets over 8 GPUs
    # cfg5 (order-4 GLL hexes, 43^3 source / 47^3 target elements) is served by gll_mesh(44, 4) / (48, 4)
}
