// A2 -- exact k nearest neighbours on the GPU.  Replaces the third-party call the reference
// makes at multi_mesh/scripts/cli.py:66-73:
//     tree = scipy.spatial.cKDTree(centroids, balanced_tree=False);  _, idx = tree.query(pts, k)
//
// Contract restated (scipy 1.15 ckdtree, p = 2, eps = 0): the k smallest Euclidean distances in
// ascending order; the squared distance is accumulated axis by axis in fp64,
// ((dx*dx + dy*dy) + dz*dz), without fused multiply-add (this file is built with
// -ffp-contract=off), so the ordering of near-equal candidates is the same as cKDTree's.
// Exactly equal distances are ordered by source index (cKDTree's tie order is traversal
// dependent and unspecified).  Rows with fewer than k sources are padded with index nsrc and
// distance inf, as cKDTree does.
//
// Structure: a uniform grid over the source bounding box replaces the k-d tree -- a counting
// sort of the sources by cell (histogram -> exclusive scan -> scatter) gives cell-contiguous
// coordinate runs.  A query lane scans the 3x3x3 block of cells around its target, then
// successive shells, keeping its k best (d2, id) pairs sorted in registers, and stops when the
// k-th best distance is closer than the nearest face of the scanned block (every unscanned
// source lies beyond that face).  Cells along z are contiguous in memory, so a block column is
// one coordinate run.
#include <math.h>

#include <new>

#include "mm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr double kTargetPerCell = 4.0;  // average sources per cell
constexpr int kMaxCellsPerAxis = 1024;

struct GridParams {
    int nx, ny, nz;
    double lox, loy, loz;
    double hx, hy, hz;
    double ihx, ihy, ihz;
};

// ---- bounding box -------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bbox_partial_kernel(const double *__restrict__ src, i64 nsrc,
                                                              int ndim, double *__restrict__ partial)
{
    __shared__ double smin[3][kBlock / 64];
    __shared__ double smax[3][kBlock / 64];
    double mn[3] = {INFINITY, INFINITY, INFINITY};
    double mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < nsrc; e += (i64)gridDim.x * blockDim.x) {
        for (int a = 0; a < ndim; ++a) {
            const double v = src[e * ndim + a];
            mn[a] = fmin(mn[a], v);
            mx[a] = fmax(mx[a], v);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fmin(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmax(mx[a], __shfl_xor(mx[a], off));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            smin[a][wave] = mn[a];
            smax[a][wave] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            double lo = smin[a][0], hi = smax[a][0];
            for (int wv = 1; wv < kBlock / 64; ++wv) {
                lo = fmin(lo, smin[a][wv]);
                hi = fmax(hi, smax[a][wv]);
            }
            partial[blockIdx.x * 6 + a] = lo;
            partial[blockIdx.x * 6 + 3 + a] = hi;
        }
    }
}

__global__ void bbox_final_kernel(const double *__restrict__ partial, int nblocks, double *__restrict__ out)
{
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        double v = partial[a];
        for (int b = 1; b < nblocks; ++b) {
            const double p = partial[b * 6 + a];
            v = a < 3 ? fmin(v, p) : fmax(v, p);
        }
        out[a] = v;
    }
}

// ---- cell assignment ----------------------------------------------------------------
__device__ __forceinline__ int cell_coord(double x, double lo, double ih, int n)
{
    double t = (x - lo) * ih;
    t = fmin(fmax(t, 0.0), (double)(n - 1));  // NaN -> 0, outside -> clamped
    return (int)t;
}

__global__ __launch_bounds__(kBlock) void cell_count_kernel(const double *__restrict__ src, i64 nsrc, int ndim,
                                                            GridParams g, int *__restrict__ cell_of,
                                                            int *__restrict__ counts)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nsrc) return;
    const double x = src[e * ndim];
    const double y = ndim > 1 ? src[e * ndim + 1] : 0.0;
    const double z = ndim > 2 ? src[e * ndim + 2] : 0.0;
    const int cx = cell_coord(x, g.lox, g.ihx, g.nx);
    const int cy = cell_coord(y, g.loy, g.ihy, g.ny);
    const int cz = cell_coord(z, g.loz, g.ihz, g.nz);
    const int c = (cx * g.ny + cy) * g.nz + cz;
    cell_of[e] = c;
    atomicAdd(&counts[c], 1);
}

// ---- exclusive scan of the per-cell counts (three small kernels) --------------------
constexpr int kScanItems = 4;                       // items per thread
constexpr int kScanTile = kBlock * kScanItems;      // items per block

__device__ __forceinline__ int block_exclusive_scan(int v, int *total)
{
    __shared__ int wave_sums[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int wv = 0; wv < kBlock / 64; ++wv) {
        if (wv < wave) base += wave_sums[wv];
        tot += wave_sums[wv];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kBlock) void scan_tile_sums_kernel(const int *__restrict__ counts, i64 n,
                                                                int *__restrict__ tile_sums)
{
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int s = 0;
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) s += counts[base + i];
    int total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void scan_tile_offsets_kernel(int *__restrict__ tile_sums, int ntiles)
{
    // single block: running exclusive scan over the tile sums
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = i < ntiles ? tile_sums[i] : 0;
        int total;
        const int excl = block_exclusive_scan(v, &total);
        const int c = carry;
        if (i < ntiles) tile_sums[i] = c + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const int *__restrict__ counts, i64 n,
                                                            const int *__restrict__ tile_offsets,
                                                            int *__restrict__ start, int *__restrict__ cursor)
{
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? counts[base + i] : 0;
        s += v[i];
    }
    int total;
    int excl = block_exclusive_scan(s, &total) + tile_offsets[blockIdx.x];
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) {
            start[base + i] = excl;
            cursor[base + i] = excl;
        }
        excl += v[i];
    }
    // start[n] = total number of items
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) start[n] = excl;
}

__global__ __launch_bounds__(kBlock) void cell_scatter_kernel(const double *__restrict__ src, i64 nsrc, int ndim,
                                                              const int *__restrict__ cell_of,
                                                              int *__restrict__ cursor,
                                                              double *__restrict__ sorted_xyz,
                                                              int *__restrict__ sorted_id)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nsrc) return;
    const int pos = atomicAdd(&cursor[cell_of[e]], 1);
    sorted_xyz[(i64)pos * 3 + 0] = src[e * ndim];
    sorted_xyz[(i64)pos * 3 + 1] = ndim > 1 ? src[e * ndim + 1] : 0.0;
    sorted_xyz[(i64)pos * 3 + 2] = ndim > 2 ? src[e * ndim + 2] : 0.0;
    sorted_id[pos] = (int)e;
}

// ---- query --------------------------------------------------------------------------
// (d2, id) lexicographic "a before b"
__device__ __forceinline__ bool before(double da, int ia, double db, int ib)
{
    return da < db || (da == db && ia < ib);
}

template <int K>
struct BestList {
    double d[K];
    int id[K];
    __device__ __forceinline__ void init(int pad_id)
    {
#pragma unroll
        for (int s = 0; s < K; ++s) {
            d[s] = INFINITY;
            id[s] = pad_id;
        }
    }
    // insert (nd, nid) keeping the list sorted; the caller has checked it beats the last slot
    __device__ __forceinline__ void insert(double nd, int nid)
    {
        double cd = nd;
        int ci = nid;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const bool lt = before(cd, ci, d[s], id[s]);
            const double td = d[s];
            const int ti = id[s];
            d[s] = lt ? cd : td;
            id[s] = lt ? ci : ti;
            cd = lt ? td : cd;
            ci = lt ? ti : ci;
        }
    }
};

template <int K>
__global__ __launch_bounds__(kBlock) void knn_query_kernel(GridParams g, i64 nsrc,
                                                           const int *__restrict__ cell_start,
                                                           const double *__restrict__ sorted_xyz,
                                                           const int *__restrict__ sorted_id,
                                                           const double *__restrict__ pts, i64 npts, int ndim,
                                                           int kout, i64 *__restrict__ idx_out,
                                                           double *__restrict__ dist_out)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const double px = pts[i * ndim];
    const double py = ndim > 1 ? pts[i * ndim + 1] : 0.0;
    const double pz = ndim > 2 ? pts[i * ndim + 2] : 0.0;
    const int cx = cell_coord(px, g.lox, g.ihx, g.nx);
    const int cy = cell_coord(py, g.loy, g.ihy, g.ny);
    const int cz = cell_coord(pz, g.loz, g.ihz, g.nz);

    BestList<K> best;
    best.init((int)nsrc);

    // slack: a source assigned to cell c may sit this far outside the cell's nominal box
    const double slack_x = 1e-9 * g.hx, slack_y = 1e-9 * g.hy, slack_z = 1e-9 * g.hz;

    int rprev = -1;  // radius already scanned completely
    for (int R = 1;; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
        for (int ix = x0; ix <= x1; ++ix) {
            const int adx = abs(ix - cx);
            for (int iy = y0; iy <= y1; ++iy) {
                const int ady = abs(iy - cy);
                const int col = (ix * g.ny + iy) * g.nz;
                // columns outside the previous block take the whole z range; inner columns only
                // the two new caps [cz-R, cz-rprev-1] and [cz+rprev+1, cz+R]
                const bool whole = max(adx, ady) > rprev;
                for (int part = 0; part < 2; ++part) {
                    int za, zb;
                    if (whole) {
                        if (part == 1) break;
                        za = z0;
                        zb = z1;
                    } else if (part == 0) {
                        za = z0;
                        zb = min(cz - rprev - 1, g.nz - 1);
                    } else {
                        za = max(cz + rprev + 1, 0);
                        zb = z1;
                    }
                    if (za > zb) continue;
                    const int s0 = cell_start[col + za];
                    const int s1 = cell_start[col + zb + 1];
                    for (int s = s0; s < s1; ++s) {
                        const double dx = sorted_xyz[(i64)s * 3 + 0] - px;
                        const double dy = sorted_xyz[(i64)s * 3 + 1] - py;
                        const double dz = sorted_xyz[(i64)s * 3 + 2] - pz;
                        double d2 = dx * dx;
                        d2 = d2 + dy * dy;
                        if (ndim > 2) d2 = d2 + dz * dz;
                        const int sid = sorted_id[s];
                        if (before(d2, sid, best.d[K - 1], best.id[K - 1])) best.insert(d2, sid);
                    }
                }
            }
        }
        rprev = R;
        // whole grid scanned?
        const bool all_x = (cx - R <= 0) && (cx + R >= g.nx - 1);
        const bool all_y = (cy - R <= 0) && (cy + R >= g.ny - 1);
        const bool all_z = (cz - R <= 0) && (cz + R >= g.nz - 1);
        if (all_x && all_y && all_z) break;
        // distance from the target to the nearest face of the scanned block that still has
        // cells behind it; every unscanned source is at least that far away (minus slack)
        double bound = INFINITY;
        if (cx - R > 0) bound = fmin(bound, (px - (g.lox + (double)(cx - R) * g.hx)) - slack_x);
        if (cx + R < g.nx - 1) bound = fmin(bound, ((g.lox + (double)(cx + R + 1) * g.hx) - px) - slack_x);
        if (cy - R > 0) bound = fmin(bound, (py - (g.loy + (double)(cy - R) * g.hy)) - slack_y);
        if (cy + R < g.ny - 1) bound = fmin(bound, ((g.loy + (double)(cy + R + 1) * g.hy) - py) - slack_y);
        if (cz - R > 0) bound = fmin(bound, (pz - (g.loz + (double)(cz - R) * g.hz)) - slack_z);
        if (cz + R < g.nz - 1) bound = fmin(bound, ((g.loz + (double)(cz + R + 1) * g.hz) - pz) - slack_z);
        // k-th best so far (kout <= K; the list keeps K, the bound needs slot kout-1)
        double kth = best.d[K - 1];
        if (kout < K) {
#pragma unroll
            for (int s = 0; s < K - 1; ++s)
                if (s == kout - 1) kth = best.d[s];
        }
        if (bound > 0.0 && kth < bound * bound) break;
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        if (s < kout) {
            idx_out[i * kout + s] = (i64)best.id[s];
            if (dist_out) dist_out[i * kout + s] = sqrt(best.d[s]);
        }
    }
}

template <int K>
void launch_query(mm_context *ctx, const mm_knn_index *ix, const GridParams &g, const double *pts, i64 npts,
                  int kout, i64 *idx, double *dist)
{
    const i64 grid = (npts + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((knn_query_kernel<K>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, g, ix->nsrc,
                       ix->cell_start, ix->sorted_xyz, ix->sorted_id, pts, npts, ix->ndim, kout, idx, dist);
}

GridParams params_of(const mm_knn_index *ix)
{
    GridParams g;
    g.nx = ix->dims[0];
    g.ny = ix->dims[1];
    g.nz = ix->dims[2];
    g.lox = ix->lo[0];
    g.loy = ix->lo[1];
    g.loz = ix->lo[2];
    g.hx = ix->h[0];
    g.hy = ix->h[1];
    g.hz = ix->h[2];
    g.ihx = ix->inv_h[0];
    g.ihy = ix->inv_h[1];
    g.ihz = ix->inv_h[2];
    return g;
}

void free_index(mm_knn_index *ix)
{
    if (!ix) return;
    if (ix->cell_start) (void)hipFree(ix->cell_start);
    if (ix->sorted_xyz) (void)hipFree(ix->sorted_xyz);
    if (ix->sorted_id) (void)hipFree(ix->sorted_id);
    delete ix;
}

}  // namespace

// Build without touching the stage timers (used by the fused pipeline too).
int mm_knn_build_impl(mm_context *ctx, const double *src_d, i64 nsrc, i64 ndim, mm_knn_index **out)
{
    *out = nullptr;
    mm_knn_index *ix = new (std::nothrow) mm_knn_index();
    if (!ix) {
        mm_set_error(MM_ERR_ALLOC, "out of host memory");
        return MM_ERR_ALLOC;
    }
    ix->nsrc = nsrc;
    ix->ndim = (int)ndim;

    // bounding box (one small synchronising readback; the build is a once-per-mesh step)
    double box[6] = {0, 0, 0, 0, 0, 0};
    if (nsrc > 0) {
        const int nblocks = (int)((nsrc + kBlock - 1) / kBlock < 1024 ? (nsrc + kBlock - 1) / kBlock : 1024);
        int rc = mm_scratch_begin(ctx, (size_t)nblocks * 6 * sizeof(double) + 6 * sizeof(double) + 1024);
        if (rc != MM_OK) { delete ix; return rc; }
        double *partial = (double *)mm_scratch_take(ctx, (size_t)nblocks * 6 * sizeof(double));
        double *d_box = (double *)mm_scratch_take(ctx, 6 * sizeof(double));
        hipLaunchKernelGGL(bbox_partial_kernel, dim3(nblocks), dim3(kBlock), 0, ctx->stream, src_d, nsrc, (int)ndim,
                           partial);
        hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, ctx->stream, partial, nblocks, d_box);
        hipError_t e = hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "bounding box: %s", hipGetErrorString(e));
            delete ix;
            return MM_ERR_HIP;
        }
    }
    // grid resolution: ~kTargetPerCell sources per cell over the axes that have extent
    double ext[3] = {0, 0, 0};
    int live = 0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = a < ndim ? box[3 + a] - box[a] : 0.0;
        if (!(ext[a] > 0.0) || !isfinite(ext[a])) ext[a] = 0.0;
        if (ext[a] > 0.0) {
            ++live;
            vol *= ext[a];
        }
    }
    const double want_cells = nsrc > 0 ? (double)nsrc / kTargetPerCell : 1.0;
    const double edge = live > 0 ? pow(vol / (want_cells > 1.0 ? want_cells : 1.0), 1.0 / live) : 1.0;
    i64 ncells = 1;
    for (int a = 0; a < 3; ++a) {
        int n = 1;
        if (ext[a] > 0.0 && edge > 0.0) {
            const double r = ceil(ext[a] / edge);
            n = r < 1.0 ? 1 : (r > kMaxCellsPerAxis ? kMaxCellsPerAxis : (int)r);
        }
        ix->dims[a] = n;
        ix->lo[a] = a < ndim && isfinite(box[a]) ? box[a] : 0.0;
        ix->h[a] = ext[a] > 0.0 ? ext[a] / n : 1.0;
        ix->inv_h[a] = 1.0 / ix->h[a];
        ncells *= n;
    }
    ix->ncells = ncells;
    const GridParams g = params_of(ix);

    hipError_t e = hipMalloc((void **)&ix->cell_start, (size_t)(ncells + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&ix->sorted_xyz, (size_t)(nsrc > 0 ? nsrc : 1) * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&ix->sorted_id, (size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int));
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_ALLOC, "kNN index allocation failed: %s", hipGetErrorString(e));
        free_index(ix);
        return MM_ERR_ALLOC;
    }
    const int ntiles = (int)((ncells + kScanTile - 1) / kScanTile);
    size_t need = mm_round256((size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int)) +     // cell_of
                  2 * mm_round256((size_t)(ncells + 1) * sizeof(int)) +          // counts, cursor
                  mm_round256((size_t)ntiles * sizeof(int)) + 4096;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) { free_index(ix); return rc; }
    int *cell_of = (int *)mm_scratch_take(ctx, (size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int));
    int *counts = (int *)mm_scratch_take(ctx, (size_t)(ncells + 1) * sizeof(int));
    int *cursor = (int *)mm_scratch_take(ctx, (size_t)(ncells + 1) * sizeof(int));
    int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
    if (!cell_of || !counts || !cursor || !tile_sums) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        free_index(ix);
        return MM_ERR_ALLOC;
    }
    e = hipMemsetAsync(counts, 0, (size_t)(ncells + 1) * sizeof(int), ctx->stream);
    if (e != hipSuccess) { mm_set_error(MM_ERR_HIP, "memset: %s", hipGetErrorString(e)); free_index(ix); return MM_ERR_HIP; }
    const unsigned gsrc = (unsigned)((nsrc + kBlock - 1) / kBlock);
    if (nsrc > 0)
        hipLaunchKernelGGL(cell_count_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, nsrc, (int)ndim, g,
                           cell_of, counts);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, ncells, tile_sums);
    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, tile_sums, ntiles);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, ncells, tile_sums,
                       ix->cell_start, cursor);
    if (nsrc > 0)
        hipLaunchKernelGGL(cell_scatter_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, nsrc, (int)ndim,
                           cell_of, cursor, ix->sorted_xyz, ix->sorted_id);
    e = hipGetLastError();
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_HIP, "kNN build launch: %s", hipGetErrorString(e));
        free_index(ix);
        return MM_ERR_HIP;
    }
    *out = ix;
    return MM_OK;
}

int mm_knn_query_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, i64 *idx_d,
                      double *dist_d)
{
    if (npts == 0 || k == 0) return MM_OK;
    const GridParams g = params_of(ix);
    const int kout = (int)k;
    if (k <= 1) launch_query<1>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 2) launch_query<2>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 4) launch_query<4>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 8) launch_query<8>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 16) launch_query<16>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 20) launch_query<20>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 25) launch_query<25>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 30) launch_query<30>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else if (k <= 40) launch_query<40>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    else launch_query<MM_KNN_MAX_K>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int mm_knn_build(mm_context *ctx, const double *src_d, int64_t nsrc, int64_t ndim, mm_knn_index **out)
{
    MM_REQUIRE(ctx != nullptr && out != nullptr, "null argument");
    MM_REQUIRE(ndim >= 1 && ndim <= 3, "ndim must be 1, 2 or 3");
    MM_REQUIRE(nsrc >= 0 && nsrc < (int64_t)0x7fffffff, "nsrc out of range");
    MM_REQUIRE(nsrc == 0 || src_d != nullptr, "null source array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    mm_stage_begin(ctx, MM_STAGE_KNN_BUILD);
    int rc = mm_knn_build_impl(ctx, src_d, nsrc, ndim, out);
    mm_stage_end(ctx, MM_STAGE_KNN_BUILD);
    return rc;
}

extern "C" int mm_knn_query(mm_context *ctx, const mm_knn_index *index, const double *pts_d, int64_t npts,
                            int64_t k, int64_t *idx_d, double *dist_d)
{
    MM_REQUIRE(ctx != nullptr && index != nullptr, "null argument");
    MM_REQUIRE(npts >= 0, "negative size");
    MM_REQUIRE(k >= 0 && k <= MM_KNN_MAX_K, "k must be in 0..MM_KNN_MAX_K");
    MM_REQUIRE(npts == 0 || k == 0 || (pts_d && idx_d), "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    mm_stage_begin(ctx, MM_STAGE_KNN_QUERY);
    int rc = mm_knn_query_impl(ctx, index, pts_d, npts, k, (i64 *)idx_d, dist_d);
    mm_stage_end(ctx, MM_STAGE_KNN_QUERY);
    return rc;
}

extern "C" void mm_knn_destroy(mm_context *ctx, mm_knn_index *index)
{
    if (!index) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    free_index(index);
}
