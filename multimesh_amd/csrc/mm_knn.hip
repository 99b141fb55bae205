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
#include <stdlib.h>

#include <new>

#include "mm_common.h"
#include <cstring>

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums);
// (mm_unique.hip: the stable LSD radix sort of 64-bit keys with 32-bit values)
size_t mm_radix_sort_scratch(i64 n);
int mm_radix_sort_pairs(mm_context *ctx, unsigned long long *ka, unsigned long long *kb, unsigned *va, unsigned *vb, i64 n,
                        int first_shift, int end_shift, void *scratch, bool *in_a);

namespace {
#include "mm_knn_grid.inc.h"
#include "mm_knn_rings.inc.h"
#include "mm_knn_tiles.inc.h"
#include "mm_knn_tree.inc.h"
#include "mm_knn_lane.inc.h"

// targets -> visiting order (counting sort by cell, same machinery as the source sort)
__global__ __launch_bounds__(kBlock) void target_scatter_kernel(const int *__restrict__ rank_of, i64 npts,
                                                                const double *__restrict__ pts, int ndim, GridParams g,
                                                                const int *__restrict__ start,
                                                                double *__restrict__ tsorted,
                                                                const int *__restrict__ list,
                                                                const int *__restrict__ list_count)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (list ? (i64)*list_count : npts)) return;
    const i64 p = list ? (i64)list[t] : t;   // the record carries the target's own index
    const double x = pts[p * ndim], y = ndim > 1 ? pts[p * ndim + 1] : 0.0, z = ndim > 2 ? pts[p * ndim + 2] : 0.0;
    const i64 pos = (i64)start[cell_of_point(x, y, z, g)] + rank_of[t];
    store_record(tsorted + pos * kRec, x, y, z, (int)p);
}

// scratch of the lane kernel's work-item prepass (knn_query_typed carves it)
constexpr int kLaneMaxK = 20;
struct LaneWork {
    int Z;
    int T, W;            // thin layers per cell layer, half-width of a target's window in thin layers
    int sorted_rows;     // rows and hand-overs by position in the cell-sorted order
    i64 nstrips_total;   // columns x strips per column
    i64 max_items;       // upper bound on the work items: strips + targets / (64 * kLaneRounds)
    int *tile_sums;      // [tiles of strips + 1]: exclusive item offsets per tile, then the number of items
    int2 *items;         // [max_items]
};

template <int K, typename IDX>
void launch_generic(mm_context *ctx, const mm_knn_index *ix, const GridParams &g, const double *pts, i64 npts,
                    int kout, IDX *idx, double *dist, const int *list, const int *list_count, int pstride = 0,
                    int list_min = -1)
{
    if (pstride == 0) pstride = ix->ndim;
    i64 grid = (npts + kBlock - 1) / kBlock;
    if (list && grid > 4096) grid = 4096;  // queue length is only known on the device: grid-stride
    hipLaunchKernelGGL((knn_query_kernel<K, IDX>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, g, ix->nsrc,
                       ix->cell_start, ix->sorted_xyz, pts, npts, ix->ndim, kout, idx, dist, list,
                       list_count, pstride, list_min, (const int *)ctx->abort_flags);
}

__global__ __launch_bounds__(kBlock) void list_all_kernel(int *__restrict__ list, int *__restrict__ count, i64 n)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[i] = (int)i;
    if (i == 0) *count = (int)n;
}

// List mode.  The length of the list is only known on the device, and the two kernels suit opposite ends: a
// handful of stragglers is a latency problem (one WAVE per target: 0.02 ms instead of 0.19 ms for the metric
// workload's 425), hundreds of thousands fill the chip one LANE per target (cfg5's shape hands over every
// target whose 20th neighbour lies outside the 27-cell window, 120 k of 7.2 M: 1.9 ms by waves -- 16 ns per
// target -- against 0.35 ms by lanes; the curves cross near 12 k targets).
// Both kernels are launched; each returns at once when the length is on the other's side of the threshold.
// K > 32: the scalar kernel only (a lane's private list must stay in registers).
static int list_wave_max()
{
    static const int v = getenv("MM_KNN_LIST_WAVE_MAX") ? atoi(getenv("MM_KNN_LIST_WAVE_MAX")) : 8192;
    return v;
}

template <int K, typename IDX>
void launch_list(mm_context *ctx, const mm_knn_index *ix, const LevelTable &lv, const double *pts, int pstride, i64 npts,
                 int kout, IDX *idx, double *dist, const int *list, const int *list_count, int keep_max)
{
    const int wave_max = K <= 32 ? list_wave_max() : -1;
    if (K <= 32 && wave_max >= 0) {
        const i64 cap = npts < wave_max ? npts : wave_max;
        const i64 grid = cap < 8192 ? (cap > 0 ? cap : 1) : 8192;
        hipLaunchKernelGGL((knn_list_wave_kernel<(K <= 32 ? K : 32), IDX>), dim3((unsigned)grid), dim3(kWave), 0,
                           ctx->stream, lv, ix->nsrc, pts, ix->ndim, pstride, kout, idx, dist, list, list_count,
                           keep_max, wave_max, (const int *)ctx->abort_flags);
    }
    if (npts <= wave_max) return;   // the list cannot be longer than the query
    i64 grid = (npts + kBlock - 1) / kBlock;
    if (grid > 4096) grid = 4096;
    if (lv.n > 1)
        hipLaunchKernelGGL((knn_query_levels_kernel<K, IDX>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream, lv,
                           ix->nsrc, pts, ix->ndim, kout, idx, dist, list, list_count, keep_max, npts, wave_max,
                           (const int *)ctx->abort_flags);
    else
        launch_generic<K, IDX>(ctx, ix, lv.g[0], pts, npts, kout, idx, dist, list, list_count, pstride, wave_max);
}

// Strips of a level that hold targets, as workgroup ids of knn_strip_kernel (nsplit parts each).
__global__ __launch_bounds__(kBlock) void strips_with_targets_kernel(GridParams g, const int *__restrict__ tstart,
                                                                     int nsplit, unsigned *__restrict__ list,
                                                                     int *__restrict__ count)
{
    const int ncols = g.nx * g.ny;
    const int cols_per_xcd = (ncols + 7) / 8;
    const int nstrips = (g.nz + kStripZ - 1) / kStripZ;
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    unsigned m = 0, xcd = 0;
    if (t < (i64)ncols * nstrips) {
        const int col = (int)(t / nstrips), strip = (int)(t - (i64)col * nstrips);
        const int cz0 = strip * kStripZ, cz1 = min(cz0 + kStripZ, g.nz);
        has = tstart[col * g.nz + cz1] > tstart[col * g.nz + cz0];
        xcd = (unsigned)(col / cols_per_xcd);
        m = (unsigned)(col - (int)xcd * cols_per_xcd) * (unsigned)nstrips + (unsigned)strip;
    }
    const unsigned long long vote = __ballot(has);
    if (!vote) return;
    const int lane = threadIdx.x & 63;
    const int first = __ffsll((long long)vote) - 1;
    int base = 0;
    if (lane == first) base = atomicAdd(count, __popcll(vote) * nsplit);
    base = __shfl(base, first);
    if (has) {
        const int at = base + __popcll(vote & ((1ull << lane) - 1ull)) * nsplit;
        for (int p = 0; p < nsplit; ++p) list[at + p] = ((m * (unsigned)nsplit + (unsigned)p) << 3) | xcd;
    }
}

template <int K, typename IDX>
void launch_fast(mm_context *ctx, const mm_knn_index *ix, const GridParams &g, const double *pts, i64 npts,
                 int kout, const int *tstart, const double *tsorted, IDX *idx, double *dist,
                 int *fb_list, int *fb_count, int *down_list, int *down_count, bool record_stage,
                 unsigned *strip_list, int *strip_count, const LaneWork *lane)
{
#ifndef MM_KNN_CAP_EXTRA_SMALL   // tuning builds only
#define MM_KNN_CAP_EXTRA_SMALL 8
#endif
    // room for the candidates of buckets jb and jb+1 beyond the k-th (fewer for short lists)
    constexpr int CAP = K + (K <= 8 ? MM_KNN_CAP_EXTRA_SMALL : 12);
    static const int dbg_stop = getenv("MM_KNN_DBG_STOP") ? atoi(getenv("MM_KNN_DBG_STOP")) : 0;
    // 8 XCD slabs of ceil(columns/8) cell columns each (see the kernel's workgroup map)
    const i64 cols = (i64)ix->dims[0] * ix->dims[1];
    const i64 cell_grid = 8 * ((cols + 7) / 8) * ix->dims[2];
    // strips along z need a grid that is deep in z; flat and 2-D grids keep the cell kernel
    static const int force = getenv("MM_KNN_KERNEL") ? (strcmp(getenv("MM_KNN_KERNEL"), "strip") == 0 ? 1 : 2) : 0;
    const bool use_strip = force == 1 || (force == 0 && ix->dims[2] >= 6);
    if (lane && K <= kLaneMaxK) {
        // round 2's kernel: work items from the strips that hold targets, then one lane per target
        constexpr int KL = K <= kLaneMaxK ? K : 1;   // (the strip-only list lengths are never instantiated)
        const int per_item = kWave * kLaneRounds;
        const int ntiles = (int)((lane->nstrips_total + kScanTile - 1) / kScanTile);
        hipLaunchKernelGGL(lane_items_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, g, tstart, lane->Z, per_item,
                           lane->nstrips_total, lane->tile_sums, lane->items, lane->max_items);
        hipLaunchKernelGGL(lane_items_offsets_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, lane->tile_sums, ntiles);
        hipLaunchKernelGGL(lane_items_place_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, g, tstart, lane->Z, per_item,
                           lane->nstrips_total, lane->tile_sums, ntiles, lane->items);
        const i64 wgs = lane->max_items;   // one workgroup per slot (a multiple of 8; < 2^31: npts and the strip count are)
        if (record_stage) mm_stage_begin(ctx, MM_STAGE_KNN_CELL);
        hipLaunchKernelGGL((knn_lane_kernel<KL, IDX>), dim3((unsigned)wgs), dim3(kWave), 0, ctx->stream, g, ix->nsrc,
                           ix->cell_start, ix->sorted_xyz, ix->ndim, kout, tstart, tsorted, idx, dist, fb_list, fb_count,
                           lane->items, (int)lane->max_items, lane->Z, per_item, lane->sorted_rows, down_list, down_count,
                           lane->T, K <= 8 ? lane->W : lane->T);   // (long lists reach farther: a full cell layer either way)
        if (record_stage) mm_stage_end(ctx, MM_STAGE_KNN_CELL);
        return;
    }
    if (record_stage) mm_stage_begin(ctx, MM_STAGE_KNN_CELL);
    if (use_strip) {
        const i64 nstrips = (ix->dims[2] + kStripZ - 1) / kStripZ;
        i64 strip_grid = 8 * ((cols + 7) / 8) * nstrips;
        // a strip's rounds are sequential in its wave: with many more targets than sources (a few
        // thousand strips of a thousand targets each) the strips are shared out, ~kSplitTargets each
        const i64 per_strip = npts / (cols * nstrips > 0 ? cols * nstrips : 1);
        i64 nsplit = (per_strip + kSplitTargets / 2) / kSplitTargets;
        static const int force_split = getenv("MM_KNN_SPLIT") ? atoi(getenv("MM_KNN_SPLIT")) : 0;
        if (force_split > 0) nsplit = force_split;
        nsplit = nsplit < 1 ? 1 : (nsplit > kMaxSplit ? kMaxSplit : nsplit);
        while (nsplit > 1 && strip_grid * nsplit > (i64)0x7fffffff) --nsplit;
        if (strip_list) {
            // a denser level: only the strips that hold targets, walked by a fixed number of workgroups
            nsplit = 1;   // (the list holds at most one entry per target)
            hipLaunchKernelGGL(strips_with_targets_kernel, dim3((unsigned)((cols * nstrips + kBlock - 1) / kBlock)),
                               dim3(kBlock), 0, ctx->stream, g, tstart, (int)nsplit, strip_list, strip_count);
            i64 walkers = (npts + kStripGroups - 1) / kStripGroups;   // no more workgroups than rounds
            if (walkers > 16384) walkers = 16384;
            hipLaunchKernelGGL((knn_strip_kernel<K, CAP, IDX, 2>), dim3((unsigned)walkers), dim3(kWave), 0, ctx->stream, g,
                               ix->nsrc, ix->cell_start, ix->sorted_xyz, ix->ndim, kout, tstart, tsorted, idx, dist,
                               fb_list, fb_count, dbg_stop, (int)nsplit, down_list, down_count, strip_list, strip_count);
        } else {
            strip_grid *= nsplit;
            if (down_list)
                hipLaunchKernelGGL((knn_strip_kernel<K, CAP, IDX, 1>), dim3((unsigned)strip_grid), dim3(kWave), 0,
                                   ctx->stream, g, ix->nsrc, ix->cell_start, ix->sorted_xyz, ix->ndim, kout, tstart,
                                   tsorted, idx, dist, fb_list, fb_count, dbg_stop, (int)nsplit, down_list, down_count,
                                   (const unsigned *)nullptr, (const int *)nullptr);
            else
                hipLaunchKernelGGL((knn_strip_kernel<K, CAP, IDX, 0>), dim3((unsigned)strip_grid), dim3(kWave), 0,
                                   ctx->stream, g, ix->nsrc, ix->cell_start, ix->sorted_xyz, ix->ndim, kout, tstart,
                                   tsorted, idx, dist, fb_list, fb_count, dbg_stop, (int)nsplit, (int *)nullptr,
                                   (int *)nullptr, (const unsigned *)nullptr, (const int *)nullptr);
        }
    } else {
        hipLaunchKernelGGL((knn_cell_kernel<K, CAP, IDX>), dim3((unsigned)cell_grid), dim3(kWave), 0, ctx->stream, g,
                           ix->nsrc, ix->cell_start, ix->sorted_xyz, pts, ix->ndim, kout, tstart, tsorted, idx, dist, fb_list, fb_count, dbg_stop);
    }
    if (record_stage) mm_stage_end(ctx, MM_STAGE_KNN_CELL);
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

void free_tree(mm_knn_tree *tr)
{
    if (!tr) return;
    if (!tr->borrowed) {
        if (tr->keys) (void)mm_raw_free(tr->keys);
        if (tr->xyz) (void)mm_raw_free(tr->xyz);
        if (tr->level) (void)mm_raw_free(tr->level);
        if (tr->coarse) (void)mm_raw_free(tr->coarse);
    }
    delete tr;
}

void free_index(mm_knn_index *ix)
{
    if (!ix) return;
    free_index(ix->fine);
    free_tree(ix->tree);
    if (!ix->borrowed) {
        if (ix->cell_start) (void)mm_raw_free(ix->cell_start);
        if (ix->sorted_xyz) (void)mm_raw_free(ix->sorted_xyz);
    }
    delete ix;
}

}  // namespace

// Exclusive prefix sum of n ints on the context's stream: start[0..n] (start[n] = total).
// tile_sums: scratch of ceil(n / 1024) ints.  Shared with the GLL locate's target ordering.
// The scan behind scan_tile_sums_kernel: with few tiles every workgroup of the last kernel adds up the sums of the tiles
// before its own by itself (at most kScanSelfTiles values, L2-resident) and the single-workgroup kernel in between -- a
// dispatch of 5 us, thirty times per mm_unique_points -- is not launched; with many tiles the three-kernel form.
// mirror_*: see scan_tile_offsets_kernel.
static void launch_scan_tail(mm_context *ctx, const int *counts, i64 n, int *tile_sums, int ntiles, int *start,
                             const long long *mirror_src = nullptr, long long *mirror_dst = nullptr, int mirror_n = 0)
{
    if (ntiles <= kScanSelfTiles) {
        hipLaunchKernelGGL(scan_apply_self_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, n, (const int *)tile_sums,
                           start, mirror_src, mirror_dst, mirror_n);
    } else {
        hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, tile_sums, ntiles, mirror_src,
                           mirror_dst, mirror_n);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, n, (const int *)tile_sums, start);
    }
}

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums)
{
    const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, n, tile_sums,
                       (unsigned long long *)nullptr, 0);
    launch_scan_tail(ctx, counts, n, tile_sums, ntiles, start);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

// One grid over the sources, laid out for ~per_cell sources per cell: counting sort of the sources by
// cell into 32-byte records.  With level_extra != null the share of sources in well-filled cells is
// read back (while the scan and the scatter are still queued, so a uniform cloud pays no idle time
// for it) and *level_extra = the number of denser levels the cloud asks for.
// The grid statistic in the pinned mirror (valid once the stream has passed the copy build_level queues): which denser
// levels the build wants -- bit l-1 set: level l (band mass = difference of the cumulative shares) -- and how many
// sources sit in sparse cells.
static int stat_verdict(const mm_context *ctx, i64 nsrc, int sample_shift, double *sparse_count)
{
    int extra = 0;
    for (int b = 0; b < kMaxLevels - 1; ++b) {
        const double above = (double)ctx->h_counters[kStatSlot + b];
        const double next = b + 1 < kMaxLevels - 1 ? (double)ctx->h_counters[kStatSlot + b + 1] : 0.0;
        if (above - next > kLevelShare * (double)nsrc) extra |= 1 << b;
    }
    *sparse_count = (double)ctx->h_counters[kStatSlot + kMaxLevels - 1] * (double)(1 << sample_shift);
    return extra;
}

// A build over a GUESSED grid (mm_knn_build_guessed): nobody waits for the bounding box or the grid statistic; the box
// is reduced into the pinned mirror on the way and *stat_shift tells mm_knn_guess_confirmed how the statistic was sampled.
struct GuessedBuild {
    const double *box_partial;
    int box_nblocks;
    int *stat_shift;
};

// The grid a box and a source count ask for: ~per_cell sources per cell, cubic cells (dims, lo, h, inv_h, ncells of *ix).
static void grid_layout(const double *box, i64 nsrc, int ndim, double per_cell, mm_knn_index *ix, int *live_axes)
{
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
    const double want_cells = nsrc > 0 ? (double)nsrc / per_cell : 1.0;
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
    if (live_axes) *live_axes = live;
}

// The counting sort of a query's targets by the cells of grid gl (count with ranks -> exclusive scan -> scatter of the
// 32-byte records), on ctx->stream.  counts must be zero.
static int sort_targets(mm_context *ctx, const GridParams &gl, i64 ncells, int ndim, const double *pts_d, i64 npts,
                        const int *list, const int *list_count, int *cell_of, int *counts, int *start, int *tile_sums,
                        double *tsorted)
{
    const unsigned gpts = (unsigned)((npts + kBlock - 1) / kBlock);
    const int ntiles = (int)((ncells + kScanTile - 1) / kScanTile);
    if (!list && ncells <= kHistCells && npts >= 64 * ncells)   // many targets over few cells (see the kernel)
        hipLaunchKernelGGL(cell_count_hist_kernel, dim3(256), dim3(kHistBlock), 0, ctx->stream, pts_d, npts, ndim, gl,
                           (int)ncells, cell_of, counts);
    else
        hipLaunchKernelGGL(cell_count_kernel, dim3(gpts), dim3(kBlock), 0, ctx->stream, pts_d, npts, ndim, gl, cell_of,
                           counts, list, list_count);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, ncells, tile_sums,
                       (unsigned long long *)nullptr, 0);
    launch_scan_tail(ctx, counts, ncells, tile_sums, ntiles, start);
    hipLaunchKernelGGL(target_scatter_kernel, dim3(gpts), dim3(kBlock), 0, ctx->stream, cell_of, npts, pts_d, ndim, gl,
                       start, tsorted, list, list_count);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

static int build_level(mm_context *ctx, const double *src_d, i64 nsrc, int ndim, const double *box, double per_cell,
                       bool use_context_buffers, int level, int *level_extra, mm_knn_index **out,
                       double *sparse_share = nullptr, bool stat_dirty = true, const GuessedBuild *guessed = nullptr)
{
    static_assert(kMaxLevels - 1 <= 8, "mm_buffer_slot reserves 8 pairs for the denser levels");
    const int slot_cells = level == 0 ? (int)MM_BUF_CELL_START : (int)MM_BUF_LEVELS + 2 * (level - 1);
    const int slot_xyz = level == 0 ? (int)MM_BUF_SORTED_XYZ : (int)MM_BUF_LEVELS + 2 * (level - 1) + 1;
    *out = nullptr;
    mm_knn_index *ix = new (std::nothrow) mm_knn_index();
    if (!ix) {
        mm_set_error(MM_ERR_ALLOC, "out of host memory");
        return MM_ERR_ALLOC;
    }
    ix->nsrc = nsrc;
    ix->ndim = ndim;
    int live = 0;
    grid_layout(box, nsrc, ndim, per_cell, ix, &live);
    const i64 ncells = ix->ncells;
    const GridParams g = params_of(ix);

    // sorted_xyz holds nsrc + 1 records: the tile staging of the strip and cell kernels lets the lane of an
    // EMPTY cell load "its first record" with the others (the value is discarded), and for the empty cells
    // behind the last source that is record nsrc -- past the end of an array of nsrc records, and past the
    // end of its mapping when nsrc * 32 bytes is a whole number of pages (a memory fault that came and went
    // with what the allocator had mapped behind it)
    hipError_t e = hipSuccess;
    if (use_context_buffers) {
        ix->borrowed = true;
        int brc = mm_buffer_get(ctx, slot_cells, (size_t)(ncells + 1) * sizeof(int), (void **)&ix->cell_start);
        if (brc == MM_OK)
            brc = mm_buffer_get(ctx, slot_xyz, (size_t)(nsrc + 1) * kRec * sizeof(double), (void **)&ix->sorted_xyz);
        if (brc != MM_OK) {
            free_index(ix);
            return brc;
        }
    } else {
        e = mm_raw_alloc(ctx->device, (void **)&ix->cell_start, (size_t)(ncells + 1) * sizeof(int));
        if (e == hipSuccess)
            e = mm_raw_alloc(ctx->device, (void **)&ix->sorted_xyz, (size_t)(nsrc + 1) * kRec * sizeof(double));
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_ALLOC, "kNN index allocation failed: %s", hipGetErrorString(e));
            free_index(ix);
            return MM_ERR_ALLOC;
        }
    }
    const int ntiles = (int)((ncells + kScanTile - 1) / kScanTile);
    size_t need = mm_round256((size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int)) +     // rank of every source in its cell
                  mm_round256((size_t)(ncells + 1) * sizeof(int)) +              // counts
                  mm_round256((size_t)ntiles * sizeof(int)) + 4096;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) { free_index(ix); return rc; }
    int *cell_of = (int *)mm_scratch_take(ctx, (size_t)(nsrc > 0 ? nsrc : 1) * sizeof(int));   // rank of every source in its cell
    int *counts = (int *)mm_scratch_take(ctx, (size_t)(ncells + 1) * sizeof(int));
    int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
    if (!cell_of || !counts || !tile_sums) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        free_index(ix);
        return MM_ERR_ALLOC;
    }
    // (whole 256-byte units -- the carve is rounded up to them --: an odd tail costs a second fill dispatch)
    rc = mm_zero_async(ctx, counts, mm_fill_span((size_t)(ncells + 1) * sizeof(int)));
    if (rc != MM_OK) { free_index(ix); return rc; }
    const unsigned gsrc = (unsigned)((nsrc + kBlock - 1) / kBlock);
    if (guessed)   // this call's own box goes to the pinned mirror, where mm_knn_guess_confirmed finds it at the end of the call
        {
            GuessBox gb6;
            for (int q = 0; q < 6; ++q) gb6.v[q] = box[q];
            hipLaunchKernelGGL(bbox_final_kernel, dim3(6), dim3(kBlock), 0, ctx->stream, guessed->box_partial, guessed->box_nblocks,
                               reinterpret_cast<double *>(ctx->h_counters + kBoxSlot),
                               reinterpret_cast<long long *>(ctx->d_counters + kStatSlot), gb6,
                               reinterpret_cast<int *>(ctx->d_counters + kMmAbortSlot));
        }
    if (nsrc > 0)
        hipLaunchKernelGGL(cell_count_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, nsrc, ndim, g, cell_of,
                           counts, (const int *)nullptr, (const int *)nullptr);
    const bool want_stat = level_extra != nullptr && nsrc >= kLevelMinSources && live > 0;
    const int sample_shift = (ncells + kBlock - 1) / kBlock >= 1024 ? 4 : 0;   // (the sparse share: see scan_tile_sums_kernel)
    if (level_extra) *level_extra = 0;
    // scan of the counts; its first kernel also accumulates the grid statistic, which is then copied to the host while
    // the rest of the scan and the scatter are still queued
    i64 *stat = ctx->d_counters + kStatSlot;
    // (slots kStatSlot .. +15 are the statistic's; bbox_final_kernel has cleared them for the
    // first grid of a build, a coarsened second one clears them here)
    if (want_stat && stat_dirty) e = hipMemsetAsync(stat, 0, 16 * sizeof(i64), ctx->stream);
    if (e == hipSuccess)
        hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, ncells, tile_sums,
                           want_stat ? (unsigned long long *)stat : (unsigned long long *)nullptr, sample_shift);
    // (the scan's second kernel writes the statistic to the pinned mirror on its way; the host waits for it -- ev_misc --
    // only after the rest of the build is queued, or not at all: mm_knn_build_guessed)
    launch_scan_tail(ctx, counts, ncells, tile_sums, ntiles, ix->cell_start,
                     want_stat ? (const long long *)stat : (const long long *)nullptr,
                     (long long *)(ctx->h_counters + kStatSlot), want_stat ? kMaxLevels : 0);
    if (want_stat) {
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_misc, ctx->stream);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "grid statistic: %s", hipGetErrorString(e));
            free_index(ix);
            return MM_ERR_HIP;
        }
    }
    if (nsrc > 0)
        hipLaunchKernelGGL(cell_scatter_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, nsrc, ndim, g, cell_of,
                           ix->cell_start, ix->sorted_xyz);
    e = hipGetLastError();
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_HIP, "kNN build launch: %s", hipGetErrorString(e));
        free_index(ix);
        return MM_ERR_HIP;
    }
    if (guessed) *guessed->stat_shift = want_stat ? sample_shift : -1;
    if (want_stat && !guessed) {
        e = hipEventSynchronize(ctx->ev_misc);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "grid statistic: %s", hipGetErrorString(e));
            free_index(ix);
            return MM_ERR_HIP;
        }
        double sparse_count = 0.0;
        *level_extra = stat_verdict(ctx, nsrc, sample_shift, &sparse_count);
        if (sparse_share) *sparse_share = sparse_count / (double)nsrc;
        static const bool dbg_build = getenv("MM_KNN_DEBUG") != nullptr;
        if (dbg_build) {
            fprintf(stderr, "[mm_knn] build: %lld cells; %% of the sources in cells of at most %d: %.1f, above", (long long)ncells,
                    kSparseCount, 100.0 * sparse_count / (double)nsrc);
            for (int b = 0; b < kMaxLevels - 1; ++b)
                fprintf(stderr, " %d: %.1f", kLevelCount[b], 100.0 * ctx->h_counters[kStatSlot + b] / (double)nsrc);
            fprintf(stderr, " -> level mask 0x%x\n", *level_extra);
        }
    }
    *out = ix;
    return MM_OK;
}

static int tree_mode();
static int tree_build(mm_context *ctx, mm_knn_index *ix, const double *src_d, i64 nsrc, const double *box, bool use_context_buffers);

// Build without touching the stage timers (used by the fused pipeline too).
int mm_knn_build_impl(mm_context *ctx, const double *src_d, i64 nsrc, i64 ndim, mm_knn_index **out,
                      bool use_context_buffers, const double *box_partial_d, int box_nblocks, bool hex8_centroids)
{
    *out = nullptr;
    // bounding box: per-workgroup boxes -- left by the fused pipeline's centroid kernel (box_partial_d), or by
    // bbox_partial_kernel here -- reduced by bbox_final_kernel straight into the context's PINNED mirror, which the
    // host reads after a stream synchronisation (no copy dispatch; the build's one mid-call wait besides the grid
    // statistic: the grid's dimensions size every launch that follows).  The same kernel clears the statistic.
    double box[6] = {0, 0, 0, 0, 0, 0};
    if (nsrc > 0) {
        double *h_box = reinterpret_cast<double *>(ctx->h_counters + kBoxSlot);
        long long *stat16 = reinterpret_cast<long long *>(ctx->d_counters + kStatSlot);
        if (box_partial_d) {
            hipLaunchKernelGGL(bbox_final_kernel, dim3(6), dim3(kBlock), 0, ctx->stream, box_partial_d, box_nblocks, h_box, stat16);
        } else {
            const int nblocks = (int)((nsrc + kBlock - 1) / kBlock < 1024 ? (nsrc + kBlock - 1) / kBlock : 1024);
            int rc = mm_scratch_begin(ctx, (size_t)nblocks * 6 * sizeof(double) + 1024);
            if (rc != MM_OK) return rc;
            double *partial = (double *)mm_scratch_take(ctx, (size_t)nblocks * 6 * sizeof(double));
            if (!partial) {
                mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
                return MM_ERR_ALLOC;
            }
            hipLaunchKernelGGL(bbox_partial_kernel, dim3(nblocks), dim3(kBlock), 0, ctx->stream, src_d, nsrc, (int)ndim, partial);
            hipLaunchKernelGGL(bbox_final_kernel, dim3(6), dim3(kBlock), 0, ctx->stream, partial, nblocks, h_box, stat16);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "bounding box: %s", hipGetErrorString(e));
            return MM_ERR_HIP;
        }
        for (int q = 0; q < 6; ++q) box[q] = h_box[q];
    }
    // grid resolution: ~per_cell sources per cell over the axes that have extent
    // (MM_KNN_PER_CELL overrides the default, for tuning experiments only)
    double per_cell = kDefaultPerCell;
    if (const char *env = getenv("MM_KNN_PER_CELL")) {
        const double v = atof(env);
        if (v >= 0.25 && v <= 4096.0) per_cell = v;
    }
    int max_levels = kMaxLevels;
    if (const char *env = getenv("MM_KNN_LEVELS")) max_levels = atoi(env) < 1 ? 1 : (atoi(env) > kMaxLevels ? kMaxLevels : atoi(env));
    int extra = 0;
    mm_knn_index *head = nullptr;
    static const double sparse_limit = getenv("MM_KNN_SPARSE_SHARE") ? atof(getenv("MM_KNN_SPARSE_SHARE")) : kSparseShare;
    int rc = MM_OK;
    double per_cell_level0 = per_cell;
    for (int scale = 1;; scale *= 2) {
        per_cell_level0 = per_cell;
        double sparse = 0.0;
        rc = build_level(ctx, src_d, nsrc, (int)ndim, box, per_cell, use_context_buffers, 0, max_levels > 1 ? &extra : nullptr,
                         &head, &sparse, /*stat_dirty=*/scale > 1 || nsrc == 0);
        if (rc != MM_OK) return rc;
        if (scale >= 4 || !(sparse > sparse_limit)) break;
        free_index(head);   // a large sparse region: cells of twice the volume
        head = nullptr;
        per_cell *= 2.0;
    }
    // a graded cloud: the adaptive index instead of a stack of denser grids (MM_KNN_TREE, see tree_mode)
    {
        const int mode = tree_mode();
        bool cube = ndim == 3 && nsrc >= kLevelMinSources;
        for (int a = 0; a < 3 && cube; ++a) cube = box[3 + a] - box[a] > 0.0 && isfinite(box[3 + a] - box[a]);
        // (box_partial_d / hex8_centroids: the fused hex8 pipeline's build, per call or for a resident source -- the sources
        // are a mesh's centroids, the queries lists of 8 and then of 20 for a few: there the tree is ahead as soon as the
        // cloud asks for levels at all)
        const bool deepest = (extra & (1 << (kMaxLevels - 2))) != 0, mesh_graded = (box_partial_d != nullptr || hex8_centroids) && extra != 0;
        if (cube && (mode == 1 || (mode == -1 && (deepest || mesh_graded) && max_levels > 1))) {
            rc = tree_build(ctx, head, src_d, nsrc, box, use_context_buffers);
            if (rc != MM_OK) {
                free_index(head);
                return rc;
            }
            extra = 0x10000;   // (no density levels; not a plain grid either: see grid_guess below)
        }
    }
    mm_knn_index *tail = head;
    for (int l = 1; l < max_levels; ++l) {
        per_cell /= kLevelRatio;
        if (!(extra & (1 << (l - 1)))) continue;
        if ((double)nsrc / per_cell > (double)kLevelMaxCells) break;
        mm_knn_index *lvl = nullptr;
        rc = build_level(ctx, src_d, nsrc, (int)ndim, box, per_cell, use_context_buffers, l, nullptr, &lvl);
        if (rc != MM_OK) {
            free_index(head);
            return rc;
        }
        tail->fine = lvl;
        tail = lvl;
    }
    *out = head;
    // the fused pipeline's next call over a source mesh of this size may start from this grid (mm_knn_build_guessed):
    // only the plain case -- one level, laid out at the default density
    if (use_context_buffers && box_partial_d && ndim == 3) {
        ctx->grid_guess.valid = !head->fine && extra == 0 && per_cell_level0 == kDefaultPerCell && max_levels > 1;
        ctx->grid_guess.nsrc = nsrc;
        for (int q = 0; q < 6; ++q) ctx->grid_guess.box[q] = box[q];
    }
    return MM_OK;
}

// The fused pipeline's build when the context remembers the grid of its previous call over a source mesh of the same
// size (ctx->grid_guess: the bounding box of the centroids; reference scripts/cli.py:183-195 and every time loop
// interpolate from ONE source mesh again and again).  The grid is laid out from the remembered box, nothing waits for
// this call's box or the grid statistic in mid-call (two host round trips with an idle GPU behind each), and
// mm_knn_guess_confirmed compares both with the guess after the call's last synchronisation -- a different box or a
// statistic that asks for density levels or a coarser grid means the call is run again the ordinary way (the kernels
// are safe on any grid: cell coordinates are clamped).  (Counting the cells inside the centroid kernel as well was
// measured: that kernel grows by what cell_count_kernel takes on its own, 68 vs 57 us -- the atomics and ranks, not
// the second read of the centroids, are its cost.)
int mm_knn_build_guessed(mm_context *ctx, const double *cen, i64 nelem, const double *box_partial, int box_nblocks,
                         mm_knn_index **out)
{
    *out = nullptr;
    GuessedBuild gb;
    gb.box_partial = box_partial;
    gb.box_nblocks = box_nblocks;
    gb.stat_shift = &ctx->grid_guess.stat_shift;
    int extra = 0;
    double sparse = 0.0;
    return build_level(ctx, cen, nelem, 3, ctx->grid_guess.box, kDefaultPerCell, true, 0, &extra, out, &sparse,
                       /*stat_dirty=*/false, &gb);
}

// After the synchronisation that ends a call built by mm_knn_build_guessed: was the guess this call's own grid?
bool mm_knn_guess_confirmed(mm_context *ctx)
{
    const double *h_box = reinterpret_cast<const double *>(ctx->h_counters + kBoxSlot);
    if (memcmp(h_box, ctx->grid_guess.box, 6 * sizeof(double)) != 0) return false;
    if (ctx->grid_guess.stat_shift < 0) return true;   // (a mesh too small for the statistic: the ordinary build skips it too)
    double sparse_count = 0.0;
    static const double sparse_limit = getenv("MM_KNN_SPARSE_SHARE") ? atof(getenv("MM_KNN_SPARSE_SHARE")) : kSparseShare;
    const int extra = stat_verdict(ctx, ctx->grid_guess.nsrc, ctx->grid_guess.stat_shift, &sparse_count);
    return extra == 0 && !(sparse_count / (double)ctx->grid_guess.nsrc > sparse_limit);
}


// ---- the density-adaptive index (mm_knn_tree.inc.h): build and query ----------------------------------------------
static TreeParams tree_params_of(const mm_knn_tree *tr)
{
    TreeParams tp;
    tp.lox = tr->lo[0];
    tp.loy = tr->lo[1];
    tp.loz = tr->lo[2];
    tp.scale = tr->scale;
    tp.size = tr->size;
    return tp;
}

// MM_KNN_TREE: 0 = never (the stack of density levels only), 1 = for every 3-D cloud of at least kLevelMinSources sources
// (tests, experiments), unset = where it is ahead: (a) where the stack ends -- the grid statistic of level 0 asks for the
// DEEPEST level the stack can lay out (sources in cells of 256 x the design density and more) --, (b) in the fused hex8
// pipeline whenever the centroids ask for density levels at all.
// Measured, round 4 (ms per pass; stack / tree; profiles/r04_graded_mesh_pipeline.json, r04_knn_graded_clouds.json): graded 10M
// hex8 mesh u^1.5 11.6 / 9.6, u^2.2 47.8 / 36.8, uniform 3.5 / 6.5; random 4M clouds, k = 20: uniform 2.4 / 16.2, u^1.5 5.9 / 15.0,
// u^2 9.9 / 16.5, u^3 25.6 / 19.7; a 27 x refined region, k = 8: 3.4 / 4.9.  Read per call.
static int tree_mode()
{
    const char *env = getenv("MM_KNN_TREE");
    return env ? (atoi(env) == 0 ? 0 : 1) : -1;
}

// The sources in Morton order over the bounding cube of `box`: keys, records, leaf levels, the search table.
static int tree_build(mm_context *ctx, mm_knn_index *ix, const double *src_d, i64 nsrc, const double *box,
                      bool use_context_buffers)
{
    mm_knn_tree *tr = new (std::nothrow) mm_knn_tree();
    if (!tr) {
        mm_set_error(MM_ERR_ALLOC, "out of host memory");
        return MM_ERR_ALLOC;
    }
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) {
        tr->lo[a] = box[a];
        if (box[3 + a] - box[a] > ext) ext = box[3 + a] - box[a];
    }
    tr->size = ext * (1.0 + 0x1p-30);
    tr->scale = (double)(1 << kTreeQ) / tr->size;
    const size_t n_sz = (size_t)nsrc;
    int rc = MM_OK;
    if (use_context_buffers) {
        tr->borrowed = true;
        rc = mm_buffer_get(ctx, MM_BUF_TREE_KEYS, n_sz * sizeof(u64), (void **)&tr->keys);
        if (rc == MM_OK) rc = mm_buffer_get(ctx, MM_BUF_TREE_XYZ, (n_sz + 1) * kRec * sizeof(double), (void **)&tr->xyz);
        if (rc == MM_OK) rc = mm_buffer_get(ctx, MM_BUF_TREE_LEVEL, n_sz, (void **)&tr->level);
        if (rc == MM_OK) rc = mm_buffer_get(ctx, MM_BUF_TREE_COARSE, (size_t)(kTreeCoarse + 2) * sizeof(int), (void **)&tr->coarse);
    } else {
        hipError_t e = mm_raw_alloc(ctx->device, (void **)&tr->keys, n_sz * sizeof(u64));
        if (e == hipSuccess) e = mm_raw_alloc(ctx->device, (void **)&tr->xyz, (n_sz + 1) * kRec * sizeof(double));
        if (e == hipSuccess) e = mm_raw_alloc(ctx->device, (void **)&tr->level, n_sz);
        if (e == hipSuccess) e = mm_raw_alloc(ctx->device, (void **)&tr->coarse, (size_t)(kTreeCoarse + 2) * sizeof(int));
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_ALLOC, "kNN tree allocation failed: %s", hipGetErrorString(e));
            rc = MM_ERR_ALLOC;
        }
    }
    if (rc != MM_OK) {
        free_tree(tr);
        return rc;
    }
    const size_t need = mm_round256(n_sz * sizeof(u64)) + 2 * mm_round256(n_sz * sizeof(unsigned)) + mm_radix_sort_scratch(nsrc) + 1024;
    rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) {
        free_tree(tr);
        return rc;
    }
    u64 *key_b = (u64 *)mm_scratch_take(ctx, n_sz * sizeof(u64));
    unsigned *val_a = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    unsigned *val_b = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    void *radix = mm_scratch_take(ctx, mm_radix_sort_scratch(nsrc));
    if (!key_b || !val_a || !val_b || !radix) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        free_tree(tr);
        return MM_ERR_ALLOC;
    }
    const TreeParams tp = tree_params_of(tr);
    const unsigned gsrc = (unsigned)((nsrc + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(tree_keys_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, nsrc, 3, 3, tp, (const int *)nullptr,
                       (const int *)nullptr, tr->keys, val_a);
    bool in_a = true;
    static_assert((kTreeBits / 8) % 2 == 0 && kTreeBits % 8 == 0, "an even number of 8-bit passes: the sorted keys end where they started");
    rc = mm_radix_sort_pairs(ctx, tr->keys, key_b, val_a, val_b, nsrc, 0, kTreeBits, radix, &in_a);
    if (rc != MM_OK || !in_a) {
        if (rc == MM_OK) mm_set_error(MM_ERR_HIP, "kNN tree: the sort ended in the wrong buffer");
        free_tree(tr);
        return rc != MM_OK ? rc : MM_ERR_HIP;
    }
    hipLaunchKernelGGL(tree_records_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, src_d, 3, 3, val_a, nsrc, (const int *)nullptr, tr->xyz);
    hipLaunchKernelGGL(tree_coarse_kernel, dim3((kTreeCoarse + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, tr->keys,
                       (int)nsrc, tr->coarse);
    hipLaunchKernelGGL(tree_leaf_level_kernel, dim3(gsrc), dim3(kBlock), 0, ctx->stream, tr->keys, nsrc, tr->level);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_HIP, "kNN tree build launch: %s", hipGetErrorString(e));
        free_tree(tr);
        return MM_ERR_HIP;
    }
    ix->tree = tr;
    return MM_OK;
}

// The single-target search of a tree query.  A lane of tree_ring_kernel walks a chain of ~10^3 dependent loads (searches on
// the keys, then the runs): 1 - 2 ms per target whatever the list's length -- hidden when hundreds of thousands of targets
// fill the chip (6 ns per target), all there is to see when the list holds a few thousand.  Lists up to kTreeWaveListMax
// therefore take one WAVE per target over the level-0 grid (knn_list_wave_kernel: no searches, 64 lanes per block of cells).
// Both kernels are launched; each returns at once when the length is on the other's side.
constexpr int kTreeWaveListMax = 65536;
template <typename IDX>
static void tree_ring(mm_context *ctx, const mm_knn_index *ix, const double *pts, int pstride, i64 npts, int kout, IDX *idx_d,
                      double *dist_d, const int *list, const int *list_count)
{
    const mm_knn_tree *tr = ix->tree;
    i64 grid = (npts + kBlock - 1) / kBlock;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    LevelTable lv;
    lv.n = 1;
    lv.g[0] = params_of(ix);
    lv.cell_start[0] = ix->cell_start;
    lv.sorted_xyz[0] = ix->sorted_xyz;
    const i64 wgrid = npts < 8192 ? (npts > 0 ? npts : 1) : 8192;
#define MM_TREE_RING(KK)                                                                                                           \
    do {                                                                                                                           \
        hipLaunchKernelGGL((knn_list_wave_kernel<KK, IDX>), dim3((unsigned)wgrid), dim3(kWave), 0, ctx->stream, lv, ix->nsrc, pts, \
                           ix->ndim, pstride, kout, idx_d, dist_d, list, list_count, kListKeepMax, kTreeWaveListMax,               \
                           (const int *)nullptr);                                                                                  \
        if (npts > kTreeWaveListMax)                                                                                               \
            hipLaunchKernelGGL((tree_ring_kernel<KK, IDX>), dim3((unsigned)grid), dim3(kBlock), 0, ctx->stream,                    \
                               tree_params_of(tr), (const u64 *)tr->keys, (const int *)tr->coarse,                                 \
                               (const unsigned char *)tr->level, (const double *)tr->xyz, (int)ix->nsrc, pts, ix->ndim, pstride,   \
                               kout, idx_d, dist_d, list, list_count, kTreeWaveListMax);                                           \
    } while (0)
    if (kout <= 1) MM_TREE_RING(1);
    else if (kout <= 2) MM_TREE_RING(2);
    else if (kout <= 4) MM_TREE_RING(4);
    else if (kout <= 8) MM_TREE_RING(8);
    else if (kout <= 16) MM_TREE_RING(16);
    else MM_TREE_RING(20);
#undef MM_TREE_RING
}

// lists shorter than this go straight to the ring search (sparse targets fill no 64-lane rounds)
constexpr i64 kTreeRingListMax = 65536;

// A query through the tree: the targets (all of them, or those of a device-side list whose length list_len the caller has
// read back) in Morton order, one work item per run of at most 256 targets of one node, the lane kernel's TREE
// instantiation, and the ring search over the level-0 grid for what it hands over.  Rows by the targets' own index, or --
// tsorted_out, no list -- by their position in the sorted order, which the caller receives.
template <typename IDX>
static int tree_query(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, int kout, IDX *idx_d,
                      double *dist_d, const double **tsorted_out, const int *list, const int *list_count, i64 list_len,
                      int pstride = 0, bool second_pass = false)
{
    // second_pass: the targets of this very query that its first windows did not serve -- they held more sources than the
    // tile (the expected load of a window assumes the node's density around it; a lattice of lines or sheets, anisotropic
    // elements, can hold several times that), or the k-th neighbour lay beyond the margin --: once more, with wider margins
    // and windows laid out for 70 % of the load.  What fails again goes to the single-target search.
    if (pstride == 0) pstride = ix->ndim;
    const mm_knn_tree *tr = ix->tree;
    const i64 n = list ? list_len : npts;
    if (n <= 0) return MM_OK;
    if (list && n < kTreeRingListMax) {
        tree_ring<IDX>(ctx, ix, pts_d, pstride, n, kout, idx_d, dist_d, list, list_count);
        MM_HIP_CHECK(hipGetLastError());
        return MM_OK;
    }
    const size_t n_sz = (size_t)n;
    static const bool unsorted_rows = getenv("MM_KNN_UNSORTED_ROWS") != nullptr;
    const bool sorted_rows = !list && tsorted_out != nullptr && !unsorted_rows;
    const int scan_tiles = (int)((n + 1 + kScanTile - 1) / kScanTile);
    const size_t need = 2 * mm_round256(n_sz * sizeof(u64)) + 2 * mm_round256(n_sz * sizeof(unsigned)) + mm_radix_sort_scratch(n) +
                        2 * mm_round256((n_sz + 1) * sizeof(int)) + mm_round256((size_t)scan_tiles * sizeof(int)) +
                        mm_round256((n_sz + 1) * sizeof(TreeItem)) + (sorted_rows ? 0 : mm_round256(n_sz * kRec * sizeof(double))) +
                        mm_round256((size_t)npts * sizeof(int)) + 256 + 4096;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) return rc;
    u64 *key_a = (u64 *)mm_scratch_take(ctx, n_sz * sizeof(u64));
    u64 *key_b = (u64 *)mm_scratch_take(ctx, n_sz * sizeof(u64));
    unsigned *val_a = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    unsigned *val_b = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    void *radix = mm_scratch_take(ctx, mm_radix_sort_scratch(n));
    int *flag = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *rank = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *scan_sums = (int *)mm_scratch_take(ctx, (size_t)scan_tiles * sizeof(int));
    TreeItem *items = (TreeItem *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(TreeItem));
    int *fb_list = (int *)mm_scratch_take(ctx, (size_t)npts * sizeof(int));
    int *fb_count = (int *)mm_scratch_take(ctx, 256);   // [0] hand-overs, [1] work items, [2] of the hand-overs: by overflow, [3] passed to the second pass
    int *down_list = nullptr;
    if (!second_pass) {
        // (outlives this pass's scratch: the second pass carves the pool anew)
        rc = mm_buffer_get(ctx, MM_BUF_TREE_DOWN, (n_sz + 64) * sizeof(int), (void **)&down_list);   // (the list, then its length)
        if (rc != MM_OK) return rc;
    }
    double *tsorted = nullptr;
    if (sorted_rows) {
        // (outlives this call's scratch: the locate stage reads it)
        rc = mm_buffer_get(ctx, MM_BUF_TSORTED, (size_t)npts * kRec * sizeof(double), (void **)&tsorted);
        if (rc != MM_OK) return rc;
    } else {
        tsorted = (double *)mm_scratch_take(ctx, n_sz * kRec * sizeof(double));
    }
    if (!key_a || !key_b || !val_a || !val_b || !radix || !flag || !rank || !scan_sums || !items || !fb_list || !fb_count || !tsorted) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        return MM_ERR_ALLOC;
    }
    if ((rc = mm_zero_async(ctx, fb_count, 256)) != MM_OK) return rc;
    const TreeParams tp = tree_params_of(tr);
    const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(tree_keys_kernel, dim3(gn), dim3(kBlock), 0, ctx->stream, pts_d, n, ix->ndim, pstride, tp, list, list_count, key_a, val_a);
    bool in_a = true;
    rc = mm_radix_sort_pairs(ctx, key_a, key_b, val_a, val_b, n, 0, kTreeBits, radix, &in_a);
    if (rc != MM_OK) return rc;
    const u64 *tkeys = in_a ? key_a : key_b;
    const unsigned *tvals = in_a ? val_a : val_b;
    u64 *node = in_a ? key_b : key_a;   // (the sort's other buffer is free now)
    hipLaunchKernelGGL(tree_records_kernel, dim3(gn), dim3(kBlock), 0, ctx->stream, pts_d, ix->ndim, pstride, tvals, n, list_count, tsorted);
    // (MM_TREE_CELL_MIN / MM_TREE_TILE_MAX: tuning experiments only)
    static const int env_cell_min = getenv("MM_TREE_CELL_MIN") ? atoi(getenv("MM_TREE_CELL_MIN")) : 0;
    static const int env_tile_max = getenv("MM_TREE_TILE_MAX") ? atoi(getenv("MM_TREE_TILE_MAX")) : 0;
    static const int env_cell_min2 = getenv("MM_TREE_CELL_MIN2") ? atoi(getenv("MM_TREE_CELL_MIN2")) : 0;
    static const int env_tile_max2 = getenv("MM_TREE_TILE_MAX2") ? atoi(getenv("MM_TREE_TILE_MAX2")) : 0;
    int cell_min = env_cell_min > 0 ? (kout <= 8 ? env_cell_min : (env_cell_min * 5 + 1) / 2) : tree_cell_min(kout);
    int tile_max = env_tile_max > 0 ? env_tile_max : kTreeTileMax;
    if (second_pass) {
        // what the first windows could not serve -- too full for the tile, or a k-th neighbour beyond the margin --: margins
        // of ~1.8 local spacings instead of ~1.3 (cells of 3 x the sources), laid out for 70 % of the tile
        cell_min = env_cell_min2 > 0 ? env_cell_min2 : (kout <= 8 ? 3 * cell_min : (3 * cell_min) / 2);
        tile_max = env_tile_max2 > 0 ? env_tile_max2 : (7 * tile_max) / 10;
    }
    hipLaunchKernelGGL(tree_target_node_kernel, dim3(gn), dim3(kBlock), 0, ctx->stream, tkeys, n, list_count, tr->keys,
                       (int)ix->nsrc, tr->coarse, tr->level, cell_min, tile_max, node);
    hipLaunchKernelGGL(tree_item_flags_kernel, dim3(gn), dim3(kBlock), 0, ctx->stream, node, n, list_count, kWave * kLaneRounds, flag);
    if ((rc = mm_exclusive_scan_int(ctx, flag, n, rank, scan_sums)) != MM_OK) return rc;
    int *nitems_d = fb_count + 1;
    hipLaunchKernelGGL(tree_items_kernel, dim3(gn), dim3(kBlock), 0, ctx->stream, node, n, list_count, flag, rank, tr->keys,
                       (int)ix->nsrc, tr->coarse, items, nitems_d);
    // the number of work items sizes the launch: one small wait (a graded cloud's pass takes milliseconds)
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 2, nitems_d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const i64 nitems = (i64) * reinterpret_cast<const int *>(ctx->h_counters + 2);
    if (nitems > 0) {
        TreeArgs ta;
        ta.tp = tp;
        ta.keys = tr->keys;
        ta.coarse = tr->coarse;
        ta.items = items;
        ta.nitems = nitems_d;
        const unsigned wgs = (unsigned)(8 * (nitems / 8 + 1));
        const GridParams g = params_of(ix);
        // thin layers per cell layer and half-width of a target's first window (a tree window has at most 6 cell layers:
        // 6 T <= kLaneThinMax).  MM_TREE_T / MM_TREE_W: tuning experiments only.
        static const int env_t = getenv("MM_TREE_T") ? atoi(getenv("MM_TREE_T")) : 0;
        static const int env_w = getenv("MM_TREE_W") ? atoi(getenv("MM_TREE_W")) : 0;
        const int T = env_t >= 1 && 6 * env_t <= kLaneThinMax ? env_t : kLaneThin;
        const int W8 = env_w >= 1 && env_w <= T ? env_w : (T == kLaneThin ? kLaneWin : T);
        if (!list) mm_stage_begin(ctx, MM_STAGE_KNN_CELL);
#define MM_TREE_LANE(KK)                                                                                                          \
    hipLaunchKernelGGL((knn_lane_kernel<KK, IDX, true>), dim3(wgs), dim3(kWave), 0, ctx->stream, g, ix->nsrc, (const int *)nullptr, \
                       (const double *)tr->xyz, ix->ndim, kout, (const int *)nullptr, (const double *)tsorted, idx_d, dist_d,      \
                       fb_list, fb_count, (const int2 *)nullptr, 0, 1, kWave * kLaneRounds, sorted_rows ? 1 : 0, down_list,        \
                       down_list ? fb_count + 3 : (int *)nullptr, T, KK <= 8 ? W8 : T, ta)
        if (kout <= 1) MM_TREE_LANE(1);
        else if (kout <= 2) MM_TREE_LANE(2);
        else if (kout <= 4) MM_TREE_LANE(4);
        else if (kout <= 8) MM_TREE_LANE(8);
        else if (kout <= 16) MM_TREE_LANE(16);
        else MM_TREE_LANE(20);
#undef MM_TREE_LANE
        if (!list) mm_stage_end(ctx, MM_STAGE_KNN_CELL);
    }
    MM_HIP_CHECK(hipGetLastError());
    static const bool dbg_query = getenv("MM_KNN_DEBUG") != nullptr;
    if (dbg_query) {
        int h[4] = {0, 0, 0, 0};
        MM_HIP_CHECK(hipMemcpyAsync(h, fb_count, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[mm_knn] tree%s: %lld targets in %lld work items, %d handed to the single-target search (%d of them by windows "
                        "too full for the tile), %d to a second pass; k = %d\n", second_pass ? " (second pass)" : "", (long long)n,
                (long long)nitems, h[0], h[2], h[3], kout);
    }
    // what the windows could not certify: the tree's own ring search (cells of the target's node size, then coarser)
    tree_ring<IDX>(ctx, ix, sorted_rows ? (const double *)tsorted : pts_d, sorted_rows ? kRec : pstride, npts, kout, idx_d, dist_d,
                   fb_list, fb_count);
    MM_HIP_CHECK(hipGetLastError());
    if (down_list && nitems > 0) {
        // windows too full for the tile: their targets once more (rows as in this pass: by sorted position or own index)
        MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 2, fb_count + 3, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const i64 n2 = (i64) * reinterpret_cast<const int *>(ctx->h_counters + 2);
        if (n2 > 0) {
            int *count2 = down_list + n_sz;   // (out of the scratch pool, with the list)
            MM_HIP_CHECK(hipMemcpyAsync(count2, fb_count + 3, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
            const int rc2 = tree_query<IDX>(ctx, ix, sorted_rows ? (const double *)tsorted : pts_d, npts, kout, idx_d, dist_d, nullptr,
                                            down_list, count2, n2, sorted_rows ? kRec : pstride, true);
            if (rc2 != MM_OK) return rc2;
        }
    }
    if (sorted_rows) *tsorted_out = tsorted;
    return MM_OK;
}

// tsorted_out (nullable): the caller can take the rows in the cell-sorted order of the targets; on return
// *tsorted_out = the sorted target records {x, y, z, index} (context buffer, valid until the next query) when the
// rows were written in that order, null when they are in the targets' own order (paths without the lane kernel).
template <typename IDX>
static int knn_query_typed(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, IDX *idx_d,
                           double *dist_d, const double **tsorted_out, const int *list0 = nullptr,
                           const int *list0_count = nullptr, i64 list0_len = -1)
{
    // list0 (device): only the targets list0[0 .. *list0_count) are served (rows by the targets' own indices as ever);
    // the tiled kernels then walk just the strips that hold any of them (k <= 32)
    if (tsorted_out) *tsorted_out = nullptr;
    if (npts == 0 || k == 0) return MM_OK;
    MM_REQUIRE(npts < (i64)0x7fffffff, "too many targets for one query");
    const GridParams g = params_of(ix);
    const int kout = (int)k;
    // a graded cloud with the adaptive index (lists the lane kernel can hold; a list only when its length is known here)
    if (ix->tree && k <= kLaneMaxK && (!list0 || list0_len >= 0) && !getenv("MM_KNN_FORCE_LIST"))
        return tree_query<IDX>(ctx, ix, pts_d, npts, kout, idx_d, dist_d, tsorted_out, list0, list0_count, list0_len);
    if (k > 32) {
        // long lists: generic ring-expansion kernel for every target (in the grid that suits it, when
        // the cloud has density levels)
        if (ix->fine) {
            LevelTable lv;
            lv.n = 0;
            for (const mm_knn_index *l = ix; l && lv.n < kMaxLevels; l = l->fine) {
                lv.g[lv.n] = params_of(l);
                lv.cell_start[lv.n] = l->cell_start;
                lv.sorted_xyz[lv.n] = l->sorted_xyz;
                ++lv.n;
            }
            const dim3 grid((unsigned)((npts + kBlock - 1) / kBlock)), block(kBlock);
            if (k <= 40)
                hipLaunchKernelGGL((knn_query_levels_kernel<40, IDX>), grid, block, 0, ctx->stream, lv, ix->nsrc, pts_d,
                                   ix->ndim, kout, idx_d, dist_d, (const int *)nullptr, (const int *)nullptr,
                                   kListKeepMax, npts, -1);
            else
                hipLaunchKernelGGL((knn_query_levels_kernel<MM_KNN_MAX_K, IDX>), grid, block, 0, ctx->stream, lv, ix->nsrc,
                                   pts_d, ix->ndim, kout, idx_d, dist_d, (const int *)nullptr, (const int *)nullptr,
                                   kListKeepMax, npts, -1);
        } else if (k <= 40) {
            launch_generic<40, IDX>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d, nullptr, nullptr);
        } else {
            launch_generic<MM_KNN_MAX_K, IDX>(ctx, ix, g, pts_d, npts, kout, idx_d, dist_d, nullptr, nullptr);
        }
        MM_HIP_CHECK(hipGetLastError());
        return MM_OK;
    }
    // One pass per density level: the level's share of the targets is counting-sorted by its cells and
    // visited strip by strip; targets whose strip is too full for the tile go down to the next level
    // (a device-side list), everything else the level cannot place goes to its generic kernel.
    int nlevels = 0;
    size_t need = 1024;
    for (const mm_knn_index *l = ix; l; l = l->fine) {
        const i64 nc = l->ncells;
        const int nt = (int)((nc + kScanTile - 1) / kScanTile);
        need += mm_round256((size_t)npts * sizeof(int)) +               // rank of every target in its cell
                (l->fine ? mm_round256((size_t)npts * sizeof(int)) : 0) +   // targets passed down
                ((l != ix || list0) ? mm_round256((size_t)npts * sizeof(unsigned)) : 0) +   // strips that hold targets
                mm_round256((size_t)npts * kRec * sizeof(double)) +     // cell-sorted target records
                2 * mm_round256((size_t)(nc + 1) * sizeof(int)) +       // counts, start
                mm_round256((size_t)nt * sizeof(int)) + 256;
        ++nlevels;
    }
    need += mm_round256((size_t)npts * sizeof(int)) + 256 * (size_t)(2 + nlevels);   // stragglers of all levels (one list), counters
    // One lane per target (knn_lane_kernel) when there is a single grid that is deep in z, the lists are
    // short and there are enough targets per cell to fill 64-lane rounds; otherwise the strip / cell kernels.
    // MM_KNN_KERNEL=lane|strip|cell forces a kernel (tuning and tests only).
    static const char *force_kernel = getenv("MM_KNN_KERNEL");
    LaneWork lane_work;
    const bool lane_base = !ix->fine && ix->dims[2] >= 6 && k <= kLaneMaxK;
    bool use_lane = lane_base && npts >= 2 * ix->ncells;
    // Targets that fill only a part of the grid -- one rank's share of a sharded target set: a slab with the full problem's
    // density inside it and nothing outside -- fail the average test above although every strip that holds targets is as
    // full as ever (round 4: a 1/8 shard of the metric's targets took 3.1 ms in the strip kernel, 0.25 ms here).  What
    // matters is targets per OCCUPIED strip, which only the device knows after the sort: the first query of a context
    // with these sizes reads the number of work items back (one small wait) and the verdict is kept for the next ones.
    bool lane_probe = false;
    if (lane_base && !use_lane && npts >= kLaneProbeMin && !getenv("MM_KNN_KERNEL")) {
        if (ctx->lane_hint.valid && ctx->lane_hint.npts == npts && ctx->lane_hint.ncells == ix->ncells) use_lane = ctx->lane_hint.dense;
        else lane_probe = use_lane = true;   // (set up as for the lane kernel; decided after the targets are sorted)
    }
    // with density levels: level 0 only (every target starts there; strips too full for the tile are passed
    // down), and only for the short lists the lane kernel is best at
    static const bool lane_level0 = !(getenv("MM_KNN_LANE_LEVEL0") && atoi(getenv("MM_KNN_LANE_LEVEL0")) == 0);
    if (!force_kernel && lane_level0 && ix->fine && ix->dims[2] >= 6 && k <= 8 && npts >= 2 * ix->ncells) use_lane = true;
    if (force_kernel) use_lane = strcmp(force_kernel, "lane") == 0 && !ix->fine && ix->dims[2] >= 2 && k <= kLaneMaxK;
    // MM_KNN_FORCE_LIST: every target through the list-mode kernel (tests of that kernel only)
    // (MM_KNN_FORCE_LIST here and MM_KNN_LEVELS / MM_KNN_PER_CELL in the build are read per call on purpose: tests switch them
    // inside one process; every other knob is read once)
    const bool force_list = getenv("MM_KNN_FORCE_LIST") != nullptr;
    if (force_list || list0) use_lane = lane_probe = false;
    static const bool unsorted_rows = getenv("MM_KNN_UNSORTED_ROWS") != nullptr;
    bool sorted_rows = use_lane && !ix->fine && tsorted_out != nullptr && !unsorted_rows;
    lane_work.sorted_rows = sorted_rows ? 1 : 0;
    if (use_lane) {
        static const int force_z = getenv("MM_KNN_LANE_Z") ? atoi(getenv("MM_KNN_LANE_Z")) : 0;
        lane_work.Z = force_z >= 1 && force_z <= kLaneZMax ? force_z : kLaneZ;
        if (lane_work.Z > ix->dims[2]) lane_work.Z = ix->dims[2];
        // thin layers: (Z + 2) T of them must fit the 64 lanes of the prefix sum; the window never reaches past one
        // cell layer (W <= T).  MM_KNN_LANE_T / MM_KNN_LANE_W: tuning experiments only.
        static const int force_t = getenv("MM_KNN_LANE_T") ? atoi(getenv("MM_KNN_LANE_T")) : 0;
        static const int force_w = getenv("MM_KNN_LANE_W") ? atoi(getenv("MM_KNN_LANE_W")) : 0;
        lane_work.T = force_t >= 1 ? force_t : kLaneThin;
        while (lane_work.T > 1 && (lane_work.Z + 2) * lane_work.T > kLaneThinMax) --lane_work.T;
        lane_work.W = force_w >= 1 ? force_w : (lane_work.T == kLaneThin ? kLaneWin : lane_work.T);
        if (lane_work.W > lane_work.T) lane_work.W = lane_work.T;
        const i64 nstrips = (ix->dims[2] + lane_work.Z - 1) / lane_work.Z;
        lane_work.nstrips_total = (i64)ix->dims[0] * ix->dims[1] * nstrips;
        // (slots: 8 per row of the permuted list, so up to 7 more than items; a multiple of 8 = the grid)
        lane_work.max_items = (lane_work.nstrips_total + npts / (kWave * kLaneRounds) + 16 + 7) / 8 * 8;
        need += mm_round256((size_t)((lane_work.nstrips_total + kScanTile) / kScanTile + 2) * sizeof(int)) +
                mm_round256((size_t)lane_work.max_items * sizeof(int2)) + 1024;
    }
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) return rc;
    int *fb_list = (int *)mm_scratch_take(ctx, (size_t)npts * sizeof(int));
    // (one block, zeroed by ONE fill: the stragglers' counter, then every level's {passed down, strips} counters)
    int *fb_count = (int *)mm_scratch_take(ctx, 256 * (size_t)(1 + nlevels));
    // (level 0's cell counts right behind the counters: ONE fill clears both)
    int *counts0 = (int *)mm_scratch_take(ctx, (size_t)(ix->ncells + 1) * sizeof(int));
    if (!fb_list || !fb_count || !counts0) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        return MM_ERR_ALLOC;
    }
    if (use_lane) {
        lane_work.tile_sums = (int *)mm_scratch_take(ctx, (size_t)((lane_work.nstrips_total + kScanTile) / kScanTile + 2) * sizeof(int));
        lane_work.items = (int2 *)mm_scratch_take(ctx, (size_t)lane_work.max_items * sizeof(int2));
        if (!lane_work.tile_sums || !lane_work.items) {
            mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
            return MM_ERR_ALLOC;
        }
    }
    const bool one_fill = !force_list && (char *)counts0 == (char *)fb_count + 256 * (size_t)(1 + nlevels);   // (not so under MM_GUARD_ALLOC)
    {
        const int zrc = mm_zero_async(ctx, fb_count, 256 * (size_t)(1 + nlevels) + (one_fill ? mm_round256((size_t)(ix->ncells + 1) * sizeof(int)) : 0));
        if (zrc != MM_OK) return zrc;
    }
    const unsigned gpts = (unsigned)((npts + kBlock - 1) / kBlock);
    const int *list = list0, *list_count = list0_count;   // level 0: every target (or the caller's list)
    int level = 0;
    if (force_list)
        hipLaunchKernelGGL(list_all_kernel, dim3(gpts), dim3(kBlock), 0, ctx->stream, fb_list, fb_count, npts);
    for (const mm_knn_index *l = ix; l && !force_list; l = l->fine, ++level) {
        const GridParams gl = params_of(l);
        const i64 ncells = l->ncells;
        const int ntiles = (int)((ncells + kScanTile - 1) / kScanTile);
        int *cell_of = (int *)mm_scratch_take(ctx, (size_t)npts * sizeof(int));   // rank of every target in its cell
        int *down_list = l->fine ? (int *)mm_scratch_take(ctx, (size_t)npts * sizeof(int)) : nullptr;
        int *counts = level == 0 ? counts0 : (int *)mm_scratch_take(ctx, (size_t)(ncells + 1) * sizeof(int));
        int *start = (int *)mm_scratch_take(ctx, (size_t)(ncells + 1) * sizeof(int));
        int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
        int *down_count = fb_count + 64 * (1 + level);
        int *strip_count = down_count + 1;
        unsigned *strip_list = (level > 0 || list0) ? (unsigned *)mm_scratch_take(ctx, (size_t)npts * sizeof(unsigned)) : nullptr;
        double *tsorted = nullptr;
        if (sorted_rows) {
            // (outlives this call's scratch: the locate stage reads it)
            int brc = mm_buffer_get(ctx, MM_BUF_TSORTED, (size_t)npts * kRec * sizeof(double), (void **)&tsorted);
            if (brc != MM_OK) return brc;
            if (!lane_probe) *tsorted_out = tsorted;
        } else {
            tsorted = (double *)mm_scratch_take(ctx, (size_t)npts * kRec * sizeof(double));
        }
        if (!tsorted || !cell_of || !counts || !start || !tile_sums || !down_count || (l->fine && !down_list) ||
            ((level > 0 || list0) && !strip_list)) {
            mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
            return MM_ERR_ALLOC;
        }
        // the targets' counting sort.  (Round 4 also ran it AHEAD of a guessed call -- its grid is known before the centroids
        // exist -- on a second stream beside the centroid kernel and the source sort: no gain, 3.51 vs 3.48 ms per step; they
        // are all bandwidth-bound and simply share the HBM.)
        if (!(level == 0 && one_fill))
            if (mm_zero_async(ctx, counts, mm_fill_span((size_t)(ncells + 1) * sizeof(int))) != MM_OK) return MM_ERR_HIP;
        {
            const int src_rc = sort_targets(ctx, gl, ncells, l->ndim, pts_d, npts, list, list_count, cell_of, counts, start, tile_sums, tsorted);
            if (src_rc != MM_OK) return src_rc;
        }
        if (lane_probe && level == 0) {
            // targets per occupied strip: the work-item count of the lane kernel's own prepass, read back once
            const int per_item = kWave * kLaneRounds;
            const int nt = (int)((lane_work.nstrips_total + kScanTile - 1) / kScanTile);
            hipLaunchKernelGGL(lane_items_sums_kernel, dim3(nt), dim3(kBlock), 0, ctx->stream, gl, start, lane_work.Z, per_item,
                               lane_work.nstrips_total, lane_work.tile_sums, lane_work.items, lane_work.max_items);
            hipLaunchKernelGGL(lane_items_offsets_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, lane_work.tile_sums, nt);
            MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 2, lane_work.tile_sums + nt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            const i64 items = (i64) * reinterpret_cast<const int *>(ctx->h_counters + 2);
            use_lane = items > 0 && npts >= kLaneProbeTargetsPerItem * items;
            ctx->lane_hint.valid = true;
            ctx->lane_hint.npts = npts;
            ctx->lane_hint.ncells = ix->ncells;
            ctx->lane_hint.dense = use_lane;
            lane_probe = false;
            sorted_rows = sorted_rows && use_lane;
            lane_work.sorted_rows = sorted_rows ? 1 : 0;
            if (sorted_rows) *tsorted_out = tsorted;
        }
#define MM_FAST(KK)                                                                                                  \
    launch_fast<KK, IDX>(ctx, l, gl, pts_d, npts, kout, start, tsorted, idx_d, dist_d, fb_list, fb_count, down_list, \
                         down_count, level == 0, strip_list, strip_count, use_lane && level == 0 ? &lane_work : nullptr)
        if (k <= 1) MM_FAST(1);
        else if (k <= 2) MM_FAST(2);
        else if (k <= 4) MM_FAST(4);
        else if (k <= 8) MM_FAST(8);
        else if (k <= 16) MM_FAST(16);
        else if (k <= 20) MM_FAST(20);
        else if (k <= 25) MM_FAST(25);
        else MM_FAST(32);
#undef MM_FAST
        MM_HIP_CHECK(hipGetLastError());
        static const bool dbg_query = getenv("MM_KNN_DEBUG") != nullptr;
        if (dbg_query) {
            int h[2] = {0, 0};
            MM_HIP_CHECK(hipMemcpyAsync(h, fb_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            MM_HIP_CHECK(hipMemcpyAsync(h + 1, down_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            fprintf(stderr, "[mm_knn] level %d of %d (%lld cells): %d targets for the generic kernel so far, %d passed down "
                            "(%lld targets in the query)\n", level, nlevels, (long long)ncells, h[0], h[1], (long long)npts);
        }
        list = down_list;
        list_count = down_count;
    }
    // what no level could place: the generic kernel, once, every target in the grid that suits its home cell
    LevelTable lv;
    lv.n = 0;
    for (const mm_knn_index *l = ix; l && lv.n < kMaxLevels; l = l->fine) {
        lv.g[lv.n] = params_of(l);
        lv.cell_start[lv.n] = l->cell_start;
        lv.sorted_xyz[lv.n] = l->sorted_xyz;
        ++lv.n;
    }
#define MM_GENERIC(KK)                                                                                               \
    launch_list<KK, IDX>(ctx, ix, lv, sorted_rows ? *tsorted_out : pts_d, sorted_rows ? kRec : ix->ndim, npts, kout,  \
                         idx_d, dist_d, fb_list, fb_count, kListKeepMax)
    if (k <= 1) MM_GENERIC(1);
    else if (k <= 2) MM_GENERIC(2);
    else if (k <= 4) MM_GENERIC(4);
    else if (k <= 8) MM_GENERIC(8);
    else if (k <= 16) MM_GENERIC(16);
    else if (k <= 20) MM_GENERIC(20);
    else if (k <= 25) MM_GENERIC(25);
    else MM_GENERIC(32);
#undef MM_GENERIC
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

int mm_knn_query_list_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k,
                           int *idx_d, const int *list, const int *list_count, i64 list_len_hint)
{
    if (npts == 0 || k == 0) return MM_OK;
    const int kout = (int)k;
    // list_len_hint >= 0: the caller has read the list's length back.  A LONG list (graded meshes: a tenth of the targets
    // exhaust their eight lazily evaluated candidates) goes through the tiled kernels level by level, restricted to the
    // strips that hold listed targets -- the list-mode kernels below serve one target per lane or per wave, which is
    // right for the stragglers of a uniform mesh and 5-10x too slow for a million targets.  (The caller must not keep
    // anything in the context's scratch pool across this call: the cascade carves it anew.)
    if ((list_len_hint >= kLongListMin || (ix->tree && list_len_hint >= 0)) && k <= 32)
        return knn_query_typed<int>(ctx, ix, pts_d, npts, k, idx_d, nullptr, nullptr, list, list_count, list_len_hint);
    LevelTable lv;
    lv.n = 0;
    for (const mm_knn_index *l = ix; l && lv.n < kMaxLevels; l = l->fine) {
        lv.g[lv.n] = params_of(l);
        lv.cell_start[lv.n] = l->cell_start;
        lv.sorted_xyz[lv.n] = l->sorted_xyz;
        ++lv.n;
    }
#define MM_LIST(KK)                                                                                                  \
    launch_list<KK, int>(ctx, ix, lv, pts_d, ix->ndim, npts, kout, idx_d, (double *)nullptr, list, list_count,       \
                         kListKeepMax)
    if (k <= 8) MM_LIST(8);
    else if (k <= 16) MM_LIST(16);
    else if (k <= 20) MM_LIST(20);
    else if (k <= 32) MM_LIST(32);
    else if (k <= 40) MM_LIST(40);
    else MM_LIST(MM_KNN_MAX_K);
#undef MM_LIST
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

// idx_is_int32: the fused pipeline keeps its candidate lists as int32 (half the bytes); the public
// mm_knn_query writes int64 like cKDTree.
int mm_knn_query_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, void *idx_d,
                      double *dist_d, bool idx_is_int32)
{
    if (idx_is_int32) return knn_query_typed<int>(ctx, ix, pts_d, npts, k, (int *)idx_d, dist_d, nullptr);
    return knn_query_typed<i64>(ctx, ix, pts_d, npts, k, (i64 *)idx_d, dist_d, nullptr);
}

// int32 rows, in the cell-sorted order of the targets when the lane kernel serves the query (see knn_query_typed)
int mm_knn_query_sorted_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, int *idx_d,
                             const double **tsorted_out)
{
    return knn_query_typed<int>(ctx, ix, pts_d, npts, k, idx_d, nullptr, tsorted_out);
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
    int rc = mm_knn_build_impl(ctx, src_d, nsrc, ndim, out, false, nullptr, 0, false);
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
    int rc = mm_knn_query_impl(ctx, index, pts_d, npts, k, idx_d, dist_d, false);
    mm_stage_end(ctx, MM_STAGE_KNN_QUERY);
    return rc;
}

#ifdef MM_LANE_STAMPS
extern "C" int mm_debug_lane_stamps(unsigned long long *out16, int reset)
{
    static unsigned long long *host = nullptr;
    const size_t bytes = (size_t)kStampSlots * 8 * sizeof(unsigned long long);
    if (!host) host = (unsigned long long *)calloc(1, bytes);
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lane_stamps), bytes) != hipSuccess) return -1;
    for (int q = 0; q < 16; ++q) out16[q] = 0;
    for (size_t b = 0; b < (size_t)kStampSlots; ++b) {
        for (int q = 0; q < 7; ++q) out16[q] += host[b * 8 + q];
        out16[15] += host[b * 8 + 7];
    }
    if (reset) {
        memset(host, 0, bytes);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stamps), host, bytes) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" void mm_knn_destroy(mm_context *ctx, mm_knn_index *index)
{
    if (!index) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    free_index(index);
}
