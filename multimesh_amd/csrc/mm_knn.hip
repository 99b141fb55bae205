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

namespace {

constexpr int kBlock = 256;
constexpr double kDefaultPerCell = 8.0;  // average sources per cell: the k = 20 ball (radius ~0.84 cell) fits the 3x3x3 block
constexpr int kMaxCellsPerAxis = 1024;
// Density levels (mm_knn_build_impl).  The tiled kernels take a target whose neighbourhood holds
// between ~0.46x and ~1.2x the density the grid was laid out for (enough sources in the 3x3x3 block
// that it contains the k-th neighbour, few enough that the strip's cells fit the tile).  A cloud whose
// density varies more than that gets further grids over the same sources, each laid out for
// kLevelRatio times the density of the one before; a target whose strip overflows the tile at one
// level is passed down to the next.  Level l (design density kLevelRatio^l times level 0's) is added
// when more than kLevelShare of the sources sit in level-0 cells holding between kLevelCount[l-1] and
// kLevelCount[l] points (the first threshold is well above what the Poisson noise of a uniform cloud
// reaches); bands without sources get no grid.
constexpr double kLevelRatio = 2.0;
constexpr int kMaxLevels = 9;
constexpr int kLevelCount[kMaxLevels - 1] = {15, 19, 38, 77, 154, 307, 614, 1229};   // ~9.6 x ratio^(l-1); 15: noise
constexpr double kLevelShare = 0.02;   // of the sources, in the band of level-0 cell counts a level serves
// The other end: when more than kSparseShare of the sources sit in level-0 cells with at most kSparseCount of them
// (a cloud with a large region at half the average density or less: there the k = 20 ball outgrows the 3x3x3
// block and the targets fall to the ring-search kernel), level 0 is rebuilt with cells of twice the volume (at
// most twice over); the denser regions then reach their cell size one level further down.
constexpr int kSparseCount = 5;
constexpr double kSparseShare = 0.25;
constexpr i64 kLevelMinSources = 4096;
constexpr i64 kLevelMaxCells = (i64)1 << 27;
constexpr int kListKeepMax = 24;   // list-mode queries: a target moves to a denser level above this home-cell count
constexpr i64 kLongListMin = MM_LONG_LIST_MIN;  // on-demand list queries at least this long go through the tiled cascade (mm_knn_query_list_impl)
constexpr int kSplitTargets = 128;   // strips with many more targets than this are shared between waves
constexpr int kMaxSplit = 64;
constexpr int kStatSlot = kMmStatSlot;  // slot of mm_context::d_counters / h_counters used for the statistic
static_assert(kMaxLevels <= 16, "the level statistic has 16 counter slots");
constexpr int kBoxSlot = kMmBoxSlot;   // six doubles of the pinned h_counters receive the sources' bounding box

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

// out: the context's PINNED mirror of its counters -- the host reads the box after a stream synchronisation, no copy
// dispatch; stat16 (nullable): the 16 words of the grid statistic the build accumulates next, cleared on the way.
// guess (mm_knn_build_guessed): the box the grid of this call was laid out from; mismatch6[a] = component a of THIS call's
// box differs from it -- the expensive kernels of a guessed call look at these six words first and return at once when
// the grid is not theirs (mm_aborted: all sources and targets sit clamped in a few boundary cells of a foreign grid, the
// ring searches would scan nearly every source for every target), the host runs the call again after its last wait.
struct GuessBox {
    double v[6];
};
__global__ __launch_bounds__(kBlock) void bbox_final_kernel(const double *__restrict__ partial, int nblocks,
                                                             double *__restrict__ out, long long *__restrict__ stat16,
                                                             GuessBox guess = GuessBox(), int *__restrict__ mismatch6 = nullptr)
{
    // one workgroup per component (grid 6): the threads stride over the per-block partials -- eight independent
    // loads in flight each for the fused pipeline's 2048 partials: ONE round trip (a single wave walking them took
    // 35 us of an otherwise idle GPU in mid-step) --, then a butterfly per wave and four values through LDS
    __shared__ double s_part[kBlock / 64];
    const int a = blockIdx.x;
    if (stat16 && a == 0 && threadIdx.x < 16) stat16[threadIdx.x] = 0;
    const double init = a < 3 ? INFINITY : -INFINITY;
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = init;
    for (int b = threadIdx.x; b < nblocks; b += 8 * kBlock) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int bb = b + kBlock * u;
            const double p = bb < nblocks ? partial[bb * 6 + a] : init;
            v[u] = a < 3 ? fmin(v[u], p) : fmax(v[u], p);
        }
    }
    double r = init;
#pragma unroll
    for (int u = 0; u < 8; ++u) r = a < 3 ? fmin(r, v[u]) : fmax(r, v[u]);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(r, off);
        r = a < 3 ? fmin(r, o) : fmax(r, o);
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kBlock / 64; ++wv) r = a < 3 ? fmin(r, s_part[wv]) : fmax(r, s_part[wv]);
        out[a] = r;
        if (mismatch6) mismatch6[a] = r == guess.v[a] ? 0 : 1;   // (NaN: a mismatch)
    }
}

// ---- cell assignment ----------------------------------------------------------------
__device__ __forceinline__ int cell_coord(double x, double lo, double ih, int n)
{
    double t = (x - lo) * ih;
    t = fmin(fmax(t, 0.0), (double)(n - 1));  // NaN -> 0, outside -> clamped
    return (int)t;
}

// The histogram atomic also hands out the item's rank inside its cell, so the scatter pass needs
// no second atomic.  Mesh-ordered points arrive in runs of equal cells (neighbours along the
// fastest axis), and same-address atomics serialise in L2: the first lane of each run of equal
// cells inside the wave adds the run's length, the others take consecutive ranks behind it.
// (Random-order input: every run has length 1, nothing lost but a dozen instructions.)
// Called by every lane of the wave (c = -1, live = false for lanes without an item).
__device__ __forceinline__ int count_and_rank(int c, bool live, int *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(c, 1);
    const bool head = lane == 0 || c != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));       // heads at lanes <= mine
    const int head_lane = 63 - __clzll((long long)upto);
    const unsigned long long after = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
    int base = 0;
    if (head && live) {
        const int next_head = after ? __ffsll((long long)after) - 1 : 64;
        base = atomicAdd(&counts[c], next_head - lane);
    }
    base = __shfl(base, head_lane);
    return base + (lane - head_lane);
}

// the cell of a point (the count and the scatter pass of a counting sort both call this: same arithmetic, same cell)
__device__ __forceinline__ int cell_of_point(double x, double y, double z, const GridParams &g)
{
    const int cx = cell_coord(x, g.lox, g.ihx, g.nx);
    const int cy = cell_coord(y, g.loy, g.ihy, g.ny);
    const int cz = cell_coord(z, g.loz, g.ihz, g.nz);
    return (cx * g.ny + cy) * g.nz + cz;
}

// With `list` the items are the points list[0 .. *list_count) (a density level's share of the targets).
// rank_of[item] = the item's rank inside its cell; the scatter pass works the cell out again from the coordinates it
// reads anyway (4 bytes per item written here and read there instead of 8: both passes move bytes, nothing else).
__global__ __launch_bounds__(kBlock) void cell_count_kernel(const double *__restrict__ src, i64 nsrc, int ndim,
                                                            GridParams g, int *__restrict__ rank_of,
                                                            int *__restrict__ counts, const int *__restrict__ list,
                                                            const int *__restrict__ list_count)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < (list ? (i64)*list_count : nsrc);
    int c = -1;
    if (live) {
        const i64 p = list ? (i64)list[e] : e;
        c = cell_of_point(src[p * ndim], ndim > 1 ? src[p * ndim + 1] : 0.0, ndim > 2 ? src[p * ndim + 2] : 0.0, g);
    }
    const int rank = count_and_rank(c, live, counts);
    if (live) rank_of[e] = rank;
}

// The same count for many items over FEW cells (the unique GLL points of a target mesh over the coarse grid of a few
// source elements: cfg5 has 7.2 M targets in 10,648 cells, and in the lexicographic order np.unique leaves them in every
// wave in flight adds to the same few dozen counters: 0.6 ms where the 10 M targets of the metric take 0.08 -- merging
// the single-cell waves of a workgroup before the add changes nothing, the contention is between workgroups).  Here a
// workgroup takes a long contiguous share of the items, counts it in an LDS histogram of the whole grid, adds every
// non-empty bin to the global counter ONCE -- the bin then holds the share's base in that cell -- and walks its share a
// second time to hand out the ranks from the bins.  Two reads of the coordinates instead of one, a few hundred global
// adds per workgroup instead of tens of thousands.
constexpr int kHistBlock = 1024;
constexpr int kHistCells = 16384;   // bins: 64 KB of LDS
__global__ __launch_bounds__(kHistBlock) void cell_count_hist_kernel(const double *__restrict__ src, i64 nsrc, int ndim,
                                                                     GridParams g, int ncells, int *__restrict__ rank_of,
                                                                     int *__restrict__ counts)
{
    __shared__ int s_bin[kHistCells];
    for (int t = threadIdx.x; t < ncells; t += kHistBlock) s_bin[t] = 0;
    __syncthreads();
    // this workgroup's share: whole chunks of kHistBlock items
    const i64 chunks = (nsrc + kHistBlock - 1) / kHistBlock;
    const i64 c_lo = chunks * blockIdx.x / gridDim.x, c_hi = chunks * (blockIdx.x + 1) / gridDim.x;
    for (i64 ch = c_lo; ch < c_hi; ++ch) {
        const i64 e = ch * kHistBlock + threadIdx.x;
        const bool live = e < nsrc;
        int c = -1;
        if (live) c = cell_of_point(src[e * ndim], ndim > 1 ? src[e * ndim + 1] : 0.0, ndim > 2 ? src[e * ndim + 2] : 0.0, g);
        (void)count_and_rank(c, live, s_bin);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ncells; t += kHistBlock) {
        const int n = s_bin[t];
        if (n > 0) s_bin[t] = atomicAdd(&counts[t], n);
    }
    __syncthreads();
    for (i64 ch = c_lo; ch < c_hi; ++ch) {
        const i64 e = ch * kHistBlock + threadIdx.x;
        const bool live = e < nsrc;
        int c = -1;
        if (live) c = cell_of_point(src[e * ndim], ndim > 1 ? src[e * ndim + 1] : 0.0, ndim > 2 ? src[e * ndim + 2] : 0.0, g);
        const int rank = count_and_rank(c, live, s_bin);   // (the bin holds base + ranks handed out so far)
        if (live) rank_of[e] = rank;
    }
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

// First kernel of the scan: per-tile sums.  With `level_total` it also accumulates the grid statistic of the build
// from the counts it reads anyway (no kernel of its own):
//   level_total[b] += the counts of the cells holding more than kLevelCount[b] sources (density levels);
//   level_total[kMaxLevels - 1] += the counts of the cells holding at most kSparseCount sources, from every
//   2^sample_shift-th run of 256 cells only (an estimate that steers a heuristic: nearly every wave of a uniform
//   cloud has such a cell, and 17 k atomics on one address are 0.1 ms).
__global__ __launch_bounds__(kBlock) void scan_tile_sums_kernel(const int *__restrict__ counts, i64 n,
                                                                int *__restrict__ tile_sums,
                                                                unsigned long long *__restrict__ level_total,
                                                                int sample_shift)
{
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? counts[base + i] : 0;
        s += v[i];
    }
    if (level_total) {
        // (a wave holds one run of 256 consecutive cells: kScanItems = 4 per lane)
        static_assert(kScanItems * 64 == 256, "the sparse share is sampled per run of 256 cells");
        const unsigned run = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if ((run & ((1u << sample_shift) - 1u)) == 0) {
            int w = 0;
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) w += v[i] <= kSparseCount ? v[i] : 0;
            if (__any(w > 0)) {
                for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
                if ((threadIdx.x & 63) == 0 && w > 0) atomicAdd(level_total + (kMaxLevels - 1), (unsigned long long)w);
            }
        }
#pragma unroll
        for (int b = 0; b < kMaxLevels - 1; ++b) {
            int w = 0;
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) w += v[i] > kLevelCount[b] ? v[i] : 0;
            if (!__any(w > 0)) break;   // thresholds ascend
            for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
            if ((threadIdx.x & 63) == 0 && w > 0) atomicAdd(level_total + b, (unsigned long long)w);
        }
    }
    int total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// mirror_src / mirror_dst (nullable): mirror_n 64-bit words copied on the way from device memory to the context's
// PINNED host mirror -- the grid statistic the kernel before this one accumulated (a copy command of the runtime's costs a
// dispatch of 4 us behind a 6 us gap)
__global__ __launch_bounds__(kBlock) void scan_tile_offsets_kernel(int *__restrict__ tile_sums, int ntiles,
                                                                    const long long *__restrict__ mirror_src = nullptr,
                                                                    long long *__restrict__ mirror_dst = nullptr,
                                                                    int mirror_n = 0)
{
    if (mirror_src && (int)threadIdx.x < mirror_n) mirror_dst[threadIdx.x] = mirror_src[threadIdx.x];
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

// scan_apply_kernel for few tiles, straight behind scan_tile_sums_kernel: the workgroup sums the tiles before its own
// itself (tile_sums are the RAW sums here)
constexpr int kScanSelfTiles = 2048;
__global__ __launch_bounds__(kBlock) void scan_apply_self_kernel(const int *__restrict__ counts, i64 n,
                                                                 const int *__restrict__ tile_sums,
                                                                 int *__restrict__ start,
                                                                 const long long *__restrict__ mirror_src,
                                                                 long long *__restrict__ mirror_dst, int mirror_n)
{
    if (mirror_src && blockIdx.x == 0 && (int)threadIdx.x < mirror_n) mirror_dst[threadIdx.x] = mirror_src[threadIdx.x];
    int before = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += kBlock) before += tile_sums[t];
    int tile_base;
    (void)block_exclusive_scan(before, &tile_base);
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? counts[base + i] : 0;
        s += v[i];
    }
    int total;
    int excl = block_exclusive_scan(s, &total) + tile_base;
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) start[base + i] = excl;
        excl += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) start[n] = excl;   // the total number of items
}

__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const int *__restrict__ counts, i64 n,
                                                            const int *__restrict__ tile_offsets,
                                                            int *__restrict__ start)
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
        }
        excl += v[i];
    }
    // start[n] = total number of items
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) start[n] = excl;
}

// Sorted records are 32 bytes {x, y, z, original index (as the bits of a double)}: an item is
// written with two 16-byte stores into its own aligned sector and read back the same way.
constexpr int kRec = 4;

__device__ __forceinline__ void store_record(double *__restrict__ rec, double x, double y, double z, int id)
{
    double2 *r2 = reinterpret_cast<double2 *>(rec);
    r2[0] = make_double2(x, y);
    r2[1] = make_double2(z, __longlong_as_double((long long)id));
}

__device__ __forceinline__ int record_id(double w) { return (int)__double_as_longlong(w); }

__global__ __launch_bounds__(kBlock) void cell_scatter_kernel(const double *__restrict__ src, i64 nsrc, int ndim,
                                                              GridParams g, const int *__restrict__ rank_of,
                                                              const int *__restrict__ start,
                                                              double *__restrict__ sorted_rec)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nsrc) return;
    const double x = src[e * ndim], y = ndim > 1 ? src[e * ndim + 1] : 0.0, z = ndim > 2 ? src[e * ndim + 2] : 0.0;
    const i64 pos = (i64)start[cell_of_point(x, y, z, g)] + rank_of[e];
    store_record(sorted_rec + pos * kRec, x, y, z, (int)e);
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

// Lower bound on the distance from the target to any source outside the (2R+1)^3 block of cells
// around (cx,cy,cz): distance to the nearest block face that still has cells behind it, minus a
// slack for sources sitting a rounding error outside their cell's nominal box.  +inf when the
// block covers the whole grid.
__device__ __forceinline__ double block_bound(const GridParams &g, double px, double py, double pz, int cx,
                                              int cy, int cz, int R)
{
    const double slack_x = 1e-9 * g.hx, slack_y = 1e-9 * g.hy, slack_z = 1e-9 * g.hz;
    double bound = INFINITY;
    if (cx - R > 0) bound = fmin(bound, (px - (g.lox + (double)(cx - R) * g.hx)) - slack_x);
    if (cx + R < g.nx - 1) bound = fmin(bound, ((g.lox + (double)(cx + R + 1) * g.hx) - px) - slack_x);
    if (cy - R > 0) bound = fmin(bound, (py - (g.loy + (double)(cy - R) * g.hy)) - slack_y);
    if (cy + R < g.ny - 1) bound = fmin(bound, ((g.loy + (double)(cy + R + 1) * g.hy) - py) - slack_y);
    if (cz - R > 0) bound = fmin(bound, (pz - (g.loz + (double)(cz - R) * g.hz)) - slack_z);
    if (cz + R < g.nz - 1) bound = fmin(bound, ((g.loz + (double)(cz + R + 1) * g.hz) - pz) - slack_z);
    return bound;
}

// ---- generic path: ring expansion with a register-resident sorted list.  Always correct for any
// density; used for the stragglers the fast kernel hands over (and for k > 32).
template <int K, typename IDX>
__device__ __forceinline__ void knn_query_one(const GridParams &g, i64 nsrc, const int *__restrict__ cell_start,
                                              const double *__restrict__ sorted_xyz,
                                              const double *__restrict__ pts,
                                              int ndim, int kout, IDX *__restrict__ idx_out,
                                              double *__restrict__ dist_out, i64 i, int pstride)
{
    // pstride: doubles per point (ndim for the caller's array, kRec for cell-sorted target records)
    const double px = pts[i * pstride];
    const double py = ndim > 1 ? pts[i * pstride + 1] : 0.0;
    const double pz = ndim > 2 ? pts[i * pstride + 2] : 0.0;
    const int cx = cell_coord(px, g.lox, g.ihx, g.nx);
    const int cy = cell_coord(py, g.loy, g.ihy, g.ny);
    const int cz = cell_coord(pz, g.loz, g.ihz, g.nz);

    BestList<K> best;
    best.init((int)nsrc);

    int rprev = -1;  // radius already scanned completely
    for (int R = 1;; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
        // k-th best so far: cells farther than that cannot contribute (only prunes once the list
        // is full, i.e. from the second ring on; equal distances are NOT pruned: ties go by index)
        double kth_now = best.d[K - 1];
        if (kout < K) {
#pragma unroll
            for (int s = 0; s < K - 1; ++s)
                if (s == kout - 1) kth_now = best.d[s];
        }
        for (int ix = x0; ix <= x1; ++ix) {
            const int adx = abs(ix - cx);
            const double cxl = g.lox + (double)ix * g.hx;
            const double ddx = fmax(fmax(cxl - px, px - (cxl + g.hx)) - 1e-9 * g.hx, 0.0);
            for (int iy = y0; iy <= y1; ++iy) {
                const int ady = abs(iy - cy);
                const double cyl = g.loy + (double)iy * g.hy;
                const double ddy = fmax(fmax(cyl - py, py - (cyl + g.hy)) - 1e-9 * g.hy, 0.0);
                const double lat2 = ddx * ddx + ddy * ddy;
                if (lat2 > kth_now) continue;
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
                    const double zl = g.loz + (double)za * g.hz, zh = g.loz + (double)(zb + 1) * g.hz;
                    const double ddz = fmax(fmax(zl - pz, pz - zh) - 1e-9 * g.hz, 0.0);
                    if (lat2 + ddz * ddz > kth_now) continue;
                    const int s0 = cell_start[col + za];
                    const int s1 = cell_start[col + zb + 1];
                    // four records per trip: their loads are in flight together (this kernel serves few,
                    // scattered targets and is bound by the latency of its dependent loads)
                    for (int s = s0; s < s1; s += 4) {
                        double2 xy[4], zw[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double2 *r2 =
                                reinterpret_cast<const double2 *>(sorted_xyz + (i64)min(s + u, s1 - 1) * kRec);
                            xy[u] = r2[0];
                            zw[u] = r2[1];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (s + u < s1) {
                                const double dx = xy[u].x - px;
                                const double dy = xy[u].y - py;
                                const double dz = zw[u].x - pz;
                                double d2 = dx * dx;
                                d2 = d2 + dy * dy;
                                if (ndim > 2) d2 = d2 + dz * dz;
                                const int sid = record_id(zw[u].y);
                                if (before(d2, sid, best.d[K - 1], best.id[K - 1])) best.insert(d2, sid);
                            }
                        }
                    }
                }
            }
        }
        rprev = R;
        const bool all_x = (cx - R <= 0) && (cx + R >= g.nx - 1);
        const bool all_y = (cy - R <= 0) && (cy + R >= g.ny - 1);
        const bool all_z = (cz - R <= 0) && (cz + R >= g.nz - 1);
        if (all_x && all_y && all_z) break;
        const double bound = block_bound(g, px, py, pz, cx, cy, cz, R);
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
            idx_out[i * kout + s] = (IDX)best.id[s];
            if (dist_out) dist_out[i * kout + s] = sqrt(best.d[s]);
        }
    }
}

template <int K, typename IDX>
__global__ __launch_bounds__(kBlock) void knn_query_kernel(GridParams g, i64 nsrc,
                                                           const int *__restrict__ cell_start,
                                                           const double *__restrict__ sorted_xyz,
                                                           const double *__restrict__ pts, i64 npts, int ndim,
                                                           int kout, IDX *__restrict__ idx_out,
                                                           double *__restrict__ dist_out,
                                                           const int *__restrict__ list,
                                                           const int *__restrict__ list_count, int pstride,
                                                           int list_min, const int *__restrict__ abort6 = nullptr)
{
    if (mm_aborted(abort6)) return;
    // list != null: only the queued targets (stragglers of the fast kernel), grid-stride
    const i64 total = list ? (i64)*list_count : npts;
    if (list && total <= list_min) return;   // short lists: knn_list_wave_kernel's
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride)
        knn_query_one<K, IDX>(g, nsrc, cell_start, sorted_xyz, pts, ndim, kout, idx_out, dist_out,
                         list ? (i64)list[q] : q, pstride);
}

// List mode over the density levels of a graded cloud: every listed target is searched in the first
// (coarsest) grid whose 3x3x3 block around the home cell holds at most 27 keep_max sources -- the kernel
// scans whole cells -- or in the last one.  One launch for all levels (one tail of slow lanes instead of one per level).
struct LevelTable {
    int n;
    GridParams g[kMaxLevels];
    const int *cell_start[kMaxLevels];
    const double *sorted_xyz[kMaxLevels];
};

template <int K, typename IDX>
__global__ __launch_bounds__(kBlock) void knn_query_levels_kernel(LevelTable lv, i64 nsrc,
                                                                  const double *__restrict__ pts, int ndim, int kout,
                                                                  IDX *__restrict__ idx_out,
                                                                  double *__restrict__ dist_out,
                                                                  const int *__restrict__ list,
                                                                  const int *__restrict__ list_count, int keep_max,
                                                                  i64 npts, int list_min,
                                                                  const int *__restrict__ abort6 = nullptr)
{
    if (mm_aborted(abort6)) return;
    const i64 total = list ? (i64)*list_count : npts;   // no list: every target (long lists, k > 32)
    if (list && total <= list_min) return;   // short lists: knn_list_wave_kernel's
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const i64 i = list ? (i64)list[q] : q;
        const double x = pts[i * ndim];
        const double y = ndim > 1 ? pts[i * ndim + 1] : 0.0;
        const double z = ndim > 2 ? pts[i * ndim + 2] : 0.0;
        int l = 0;
        for (; l < lv.n - 1; ++l) {
            // sources in the 3x3x3 block around the home cell (nine runs along z): what the search
            // scans at least
            const GridParams &g = lv.g[l];
            const int cx = cell_coord(x, g.lox, g.ihx, g.nx), cy = cell_coord(y, g.loy, g.ihy, g.ny);
            const int cz = cell_coord(z, g.loz, g.ihz, g.nz);
            const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.nz - 1);
            int block = 0;
            for (int ix = max(cx - 1, 0); ix <= min(cx + 1, g.nx - 1); ++ix)
                for (int iy = max(cy - 1, 0); iy <= min(cy + 1, g.ny - 1); ++iy) {
                    const int c = (ix * g.ny + iy) * g.nz;
                    block += lv.cell_start[l][c + z1 + 1] - lv.cell_start[l][c + z0];
                }
            if (block <= 27 * keep_max) break;
        }
        knn_query_one<K, IDX>(lv.g[l], nsrc, lv.cell_start[l], lv.sorted_xyz[l], pts, ndim, kout, idx_out, dist_out, i, ndim);
    }
}

// ---- fast path: one wave per grid cell, sources staged in LDS --------------------------------
// A wave owns one cell of the search grid and serves every target that falls into it.
//   stage : the cell's 3x3x3 neighbourhood (9 column runs of the cell-sorted source array) is read
//           ONCE with coalesced loads and kept in LDS as float4 {x,y,z relative to the cell corner,
//           position in the sorted array} -- instead of every lane chasing its own candidates
//           through L1 (~20 cache-line lookups per divergent load).
//   split : the 64 lanes form groups of S lanes per target (S chosen so that one round covers the
//           cell's targets, 8 <= S <= 64); lane `sl` of a group handles tile entries sl, sl+S, ...
//           Lanes of different groups read the same tile address (LDS broadcast).
//   P1    : fp32 squared distances (fused multiply-adds: this pass is only a filter) binned into a
//           64-bucket histogram per target (LDS atomics; the bucket range comes from the
//           neighbourhood's source density); each lane also keeps its candidates' bucket numbers
//           packed in registers.  jb = first bucket whose running count reaches k.
//   P2    : every candidate in a bucket <= jb+1 is appended to the target's list -- a superset of
//           the exact k nearest including exact ties (see the error bound) -- without touching the
//           distances again.
//   exact : for the ~k listed candidates only, d2 in fp64 exactly as the reference computes it
//           (coordinates re-read from the fp64 source array) and the source id.
//   P3    : rank sort of the list by exact d2 (ties: a second, lexicographic (d2, id) pass that
//           only runs when two listed distances are bit-equal); rank r < k goes to output slot r.
// Error bound.  Tile and target coordinates are rounded to fp32 relative to the cell corner O, so a
// coordinate difference is off by at most u(|s-O| + |p-O|) + u|diff| per axis (u = 2^-24) and the
// fp32 distance d32 differs from the exact distance d by at most E + 2u*d with
// E = 3u * sum_axes(|p-O| + 2h).  At least k candidates have a fp32 squared distance below the
// upper edge e1 of bucket jb, so the exact k-th distance is <= D = sqrt(e1)(1+4u) + E, and every
// candidate at exact distance <= D has d32 <= D(1+4u) + E.  The kernel checks that this is below
// the upper edge of bucket jb+1 (true unless the buckets are absurdly narrow), which makes
// "bucket <= jb+1" a superset of the exact k nearest.
// A target is handed to the generic kernel (queue) when the neighbourhood holds fewer than k
// sources, more than the tile or a column run longer than 64, the k-th distance falls outside the
// histogram range, its list overflows (many exact ties), or the exact k-th distance is not closer
// than the nearest block face (a nearer source could sit outside the block).
constexpr int kWave = 64;
constexpr int kHistBuckets = 64;
constexpr int kTileCap = 256;       // sources per tile (27 cells x ~8 expected)
constexpr int kMaxGroups = 8;       // targets per round at the narrowest split (S = 8)
constexpr int kSlots = kTileCap / kMaxGroups;  // tile entries per lane at the narrowest split

// inclusive prefix sum inside groups of S consecutive lanes (S a power of two)
__device__ __forceinline__ int group_scan(int v, int sl, int S)
{
    for (int d = 1; d < S; d <<= 1) {
        const int t = __shfl_up(v, d, S);
        if (sl >= d) v += t;
    }
    return v;
}

// The fast kernel's workgroup is ONE wave: its LDS accesses are served in program order by the LDS
// queue, so a later read sees an earlier write/atomic of any lane without waiting or s_barrier.  All
// that is needed is to keep the COMPILER from moving LDS accesses across the hand-over points.
// (__syncthreads() would also drain every outstanding global load -- s_waitcnt vmcnt(0) -- at each
// of the seven points per round, exposing the full memory latency each time.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- list mode, one WAVE per target ----------------------------------------------------------
// The targets the fast kernels hand over are few (425 of 10M on the metric workload) and scattered; one lane
// per target walking its rings record by record is a chain of ~250 dependent loads (0.19 ms for those 425
// targets, a tail nothing overlaps).  Here the 64 lanes of a wave share one target: per ring
//   runs   : lane j takes column j of the (2R+1)^2 block and looks up the extents of its one or two z-runs
//            (the whole column outside the previous block, the two caps inside it), pruned like the scalar
//            kernel by the k-th distance so far -- ONE round trip for the whole ring;
//   scan   : the runs' records are numbered through by a prefix sum (offsets in LDS) and record t goes to
//            lane t mod 64 (binary search over the <= 128 offsets); every lane keeps a private sorted list
//            of the K best of ITS records, ordered by (d2, id) -- the union of the lists holds the K best;
//   merge  : kout rounds of a wave-wide minimum over the lanes' list heads give the merged order (lane s
//            keeps entry s) and the k-th distance for the stop test.
// The set of records scanned is a superset of the scalar kernel's at every ring and the order (d2, id) is
// total, so the result is the same list, bit for bit.
constexpr int kWaveRuns = 2 * kWave;

template <int K, typename IDX>
__device__ __forceinline__ void knn_query_wave(const GridParams &g, i64 nsrc, const int *__restrict__ cell_start,
                                               const double *__restrict__ sorted_xyz, double px, double py,
                                               double pz, int ndim, int kout, IDX *__restrict__ idx_row,
                                               double *__restrict__ dist_row, int *s_off, int *s_beg)
{
    const int lane = threadIdx.x;
    const int cx = cell_coord(px, g.lox, g.ihx, g.nx);
    const int cy = cell_coord(py, g.loy, g.ihy, g.ny);
    const int cz = cell_coord(pz, g.loz, g.ihz, g.nz);
    BestList<K> best;
    best.init((int)nsrc);
    double merged_d = INFINITY;   // lane s: entry s of the merged list
    int merged_id = (int)nsrc;
    double kth = INFINITY;        // its entry kout-1 (uniform)
    int rprev = -1;
    for (int R = 1;; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
        const int ncy = y1 - y0 + 1, ncols = (x1 - x0 + 1) * ncy;
        for (int c0 = 0; c0 < ncols; c0 += kWave) {
            // ---- runs of this chunk of columns
            const int j = c0 + lane;
            int beg[2] = {0, 0}, cnt[2] = {0, 0};
            if (j < ncols) {
                const int ix = x0 + j / ncy, iy = y0 + j % ncy;
                const double cxl = g.lox + (double)ix * g.hx;
                const double ddx = fmax(fmax(cxl - px, px - (cxl + g.hx)) - 1e-9 * g.hx, 0.0);
                const double cyl = g.loy + (double)iy * g.hy;
                const double ddy = fmax(fmax(cyl - py, py - (cyl + g.hy)) - 1e-9 * g.hy, 0.0);
                const double lat2 = ddx * ddx + ddy * ddy;
                const bool whole = max(abs(ix - cx), abs(iy - cy)) > rprev;
                const int col = (ix * g.ny + iy) * g.nz;
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    int za, zb;
                    if (whole) {
                        za = z0;
                        zb = part == 0 ? z1 : z0 - 1;
                    } else if (part == 0) {
                        za = z0;
                        zb = min(cz - rprev - 1, g.nz - 1);
                    } else {
                        za = max(cz + rprev + 1, 0);
                        zb = z1;
                    }
                    if (za > zb || lat2 > kth) continue;
                    const double zl = g.loz + (double)za * g.hz, zh = g.loz + (double)(zb + 1) * g.hz;
                    const double ddz = fmax(fmax(zl - pz, pz - zh) - 1e-9 * g.hz, 0.0);
                    if (lat2 + ddz * ddz > kth) continue;
                    beg[part] = cell_start[col + za];
                    cnt[part] = cell_start[col + zb + 1] - beg[part];
                }
            }
            const int incl = group_scan(cnt[0] + cnt[1], lane, kWave);
            const int total = __shfl(incl, kWave - 1);
            const int excl = incl - cnt[0] - cnt[1];
            s_off[2 * lane] = excl;
            s_off[2 * lane + 1] = excl + cnt[0];
            s_beg[2 * lane] = beg[0];
            s_beg[2 * lane + 1] = beg[1];
            wave_sync();
            // ---- the runs' records, one per lane and trip
            for (int t0 = 0; t0 < total; t0 += kWave) {
                const int t = t0 + lane;
                int slot = 0;
#pragma unroll
                for (int step = kWaveRuns / 2; step >= 1; step >>= 1)
                    if (s_off[slot + step] <= t) slot += step;   // last run starting at or before t (empty runs share offsets)
                const bool active = t < total;
                const i64 rec = active ? (i64)s_beg[slot] + (t - s_off[slot]) : 0;
                const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + rec * kRec);
                const double2 xy = r2[0], zw = r2[1];
                const double dx = xy.x - px;
                const double dy = xy.y - py;
                const double dz = zw.x - pz;
                double d2 = dx * dx;
                d2 = d2 + dy * dy;
                if (ndim > 2) d2 = d2 + dz * dz;
                const int sid = record_id(zw.y);
                if (active && before(d2, sid, best.d[K - 1], best.id[K - 1])) best.insert(d2, sid);
            }
            wave_sync();   // the offsets are rewritten by the next chunk
        }
        // ---- merged order of the lanes' lists
        int head = 0;
        for (int s = 0; s < kout; ++s) {
            double hd = INFINITY;
            int hi = (int)nsrc;
#pragma unroll
            for (int u = 0; u < K; ++u)
                if (u == head) {
                    hd = best.d[u];
                    hi = best.id[u];
                }
            double wd = hd;
            int wi = hi;
#pragma unroll
            for (int off = kWave / 2; off >= 1; off >>= 1) {
                const double od = __shfl_xor(wd, off);
                const int oi = __shfl_xor(wi, off);
                const bool lt = before(od, oi, wd, wi);
                wd = lt ? od : wd;
                wi = lt ? oi : wi;
            }
            if (hd == wd && hi == wi && head < K) ++head;   // ids are unique: one lane gives up its head (pads: any)
            if (lane == s) {
                merged_d = wd;
                merged_id = wi;
            }
            kth = wd;
        }
        rprev = R;
        const bool all_x = (cx - R <= 0) && (cx + R >= g.nx - 1);
        const bool all_y = (cy - R <= 0) && (cy + R >= g.ny - 1);
        const bool all_z = (cz - R <= 0) && (cz + R >= g.nz - 1);
        if (all_x && all_y && all_z) break;
        const double bound = block_bound(g, px, py, pz, cx, cy, cz, R);
        if (bound > 0.0 && kth < bound * bound) break;
    }
    if (lane < kout) {
        idx_row[lane] = (IDX)merged_id;
        if (dist_row) dist_row[lane] = sqrt(merged_d);
    }
}

template <int K, typename IDX>
__global__ __launch_bounds__(kWave) void knn_list_wave_kernel(LevelTable lv, i64 nsrc,
                                                              const double *__restrict__ pts, int ndim, int pstride,
                                                              int kout, IDX *__restrict__ idx_out,
                                                              double *__restrict__ dist_out,
                                                              const int *__restrict__ list,
                                                              const int *__restrict__ list_count, int keep_max,
                                                              int list_max, const int *__restrict__ abort6 = nullptr)
{
    __shared__ int s_off[kWaveRuns];
    __shared__ int s_beg[kWaveRuns];
    if (mm_aborted(abort6)) return;
    const int total = *list_count;
    if (total > list_max) return;   // long lists fill the chip one lane per target: the scalar kernels'
    for (int q = blockIdx.x; q < total; q += gridDim.x) {
        const i64 i = list[q];
        const double x = pts[i * pstride];
        const double y = ndim > 1 ? pts[i * pstride + 1] : 0.0;
        const double z = ndim > 2 ? pts[i * pstride + 2] : 0.0;
        int l = 0;
        for (; l < lv.n - 1; ++l) {   // (the level rule of knn_query_levels_kernel)
            const GridParams &g = lv.g[l];
            const int cx = cell_coord(x, g.lox, g.ihx, g.nx), cy = cell_coord(y, g.loy, g.ihy, g.ny);
            const int cz = cell_coord(z, g.loz, g.ihz, g.nz);
            const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.nz - 1);
            int block = 0;
            for (int ix = max(cx - 1, 0); ix <= min(cx + 1, g.nx - 1); ++ix)
                for (int iy = max(cy - 1, 0); iy <= min(cy + 1, g.ny - 1); ++iy) {
                    const int c = (ix * g.ny + iy) * g.nz;
                    block += lv.cell_start[l][c + z1 + 1] - lv.cell_start[l][c + z0];
                }
            if (block <= 27 * keep_max) break;
        }
        knn_query_wave<K, IDX>(lv.g[l], nsrc, lv.cell_start[l], lv.sorted_xyz[l], x, y, z, ndim, kout,
                               idx_out + i * kout, dist_out ? dist_out + i * kout : nullptr, s_off, s_beg);
    }
}

template <int K, int CAP, typename IDX>
__global__ __launch_bounds__(kWave, 4) void knn_cell_kernel(GridParams g, i64 nsrc,
                                                            const int *__restrict__ cell_start,
                                                            const double *__restrict__ sorted_xyz,
                                                             const double *__restrict__ pts, int ndim, int kout,
                                                            const int *__restrict__ tstart,
                                                            const double *__restrict__ tsorted,
                                                            IDX *__restrict__ idx_out,
                                                            double *__restrict__ dist_out,
                                                            int *__restrict__ fb_list, int *__restrict__ fb_count,
                                                            int dbg_stop)
{
    static_assert(CAP <= 64, "rank mask is 64 bits");
    __shared__ float4 tile[kTileCap + 1];                       // +1: far-away sentinel entry
    __shared__ unsigned s_pk[kSlots / 4][kWave];                // bucket numbers of each lane's slots
    __shared__ double s_bd[CAP][kMaxGroups];
    __shared__ int s_bx[CAP][kMaxGroups];                       // source position, then source id
    __shared__ unsigned s_hist[kHistBuckets + 1][kMaxGroups];   // last row: sink for idle lanes
    __shared__ int s_jb[kMaxGroups];
    __shared__ int s_cnt[kMaxGroups];
    __shared__ unsigned long long s_seen[kMaxGroups];

    const int lane = threadIdx.x;
    // XCD-aware cell -> workgroup map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b
    // and b+8 share one), and each XCD has a private 4 MiB L2.  A cell's 27-cell neighbourhood
    // overlaps its neighbours', so every source is staged ~27 times: with the natural order those
    // re-reads land on different XCDs and only the Infinity Cache catches them (measured: ~8x the
    // algorithmic bytes crossed the fabric).  Here XCD x owns a contiguous slab of (cx,cy) columns,
    // walked z-fastest, so a source's re-reads come from the same L2.  Speed only: any placement
    // gives the same result.
    const int ncols = g.nx * g.ny;
    const int cols_per_xcd = (ncols + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const int m = blockIdx.x >> 3;
    const int col = xcd * cols_per_xcd + m / g.nz;
    if (m / g.nz >= cols_per_xcd || col >= ncols) return;
    const int cz = m % g.nz;
    const int cx = col / g.ny, cy = col % g.ny;
    const int cell = col * g.nz + cz;

    // metadata: the cell's target range and the 9 column runs of its neighbourhood, all loads
    // issued together (every address depends on the block index only)
    const int t0 = tstart[cell];
    const int t1 = tstart[cell + 1];
    const int za = max(cz - 1, 0), zb = min(cz + 1, g.nz - 1);
    int rs[9], rl[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int ix = cx + (c / 3) - 1, iy = cy + (c % 3) - 1;
        const bool inside = (unsigned)ix < (unsigned)g.nx && (unsigned)iy < (unsigned)g.ny;
        const int col = inside ? (ix * g.ny + iy) * g.nz : 0;
        const int s0 = cell_start[col + za];
        const int s1 = cell_start[col + zb + 1];
        rs[c] = inside ? s0 : 0;
        rl[c] = inside ? s1 - s0 : 0;
    }
    const int tn = t1 - t0;
    if (tn == 0) return;
    if (dbg_stop == 6) return;
    int total = 0;
    bool runs_fit = true;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        total += rl[c];
        runs_fit = runs_fit && rl[c] <= kWave;
    }
    const double ox = g.lox + (double)cx * g.hx;
    const double oy = g.loy + (double)cy * g.hy;
    const double oz = g.loz + (double)cz * g.hz;

    // histogram range from the local density: the ball holding k of the block's `total` sources
    // has r^d = (k/total) * V_block / c_d; buckets are uniform in r^2 over [0, 2.2 r^2).  Only a
    // heuristic range, so fast exp2/log2 are fine.
    float scale;
    {
        const int bx = min(cx + 1, g.nx - 1) - max(cx - 1, 0) + 1;
        const int by = min(cy + 1, g.ny - 1) - max(cy - 1, 0) + 1;
        const int bz = zb - za + 1;
        int d = 0;
        float vol = 1.f;
        if (g.nx > 1) { ++d; vol *= (float)bx * (float)g.hx; }
        if (g.ny > 1) { ++d; vol *= (float)by * (float)g.hy; }
        if (g.nz > 1) { ++d; vol *= (float)bz * (float)g.hz; }
        const float frac = (float)kout / (float)max(total, 1);
        float r2;
        if (d == 3) r2 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(frac * vol * (1.f / 4.18879f)) * (2.f / 3.f));
        else if (d == 2) r2 = frac * vol * (1.f / 3.14159f);
        else if (d == 1) { const float r = frac * vol * 0.5f; r2 = r * r; }
        else r2 = 1.f;
        scale = (float)kHistBuckets / (2.2f * r2);
    }
    const bool cell_ok = runs_fit && total >= kout && total <= kTileCap && scale > 0.f && scale < INFINITY;
    if (dbg_stop == 7) { if (total == 12345 && scale == 1.f) fb_list[0] = 1; return; }
    if (!cell_ok) {
        // the whole cell goes to the generic kernel
        for (int q = lane; q < tn; q += kWave)
            fb_list[atomicAdd(fb_count, 1)] = record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
        return;
    }

    // lanes per target: the widest split whose round still covers all of the cell's targets
    int S = kWave;
    while (S > kWave / kMaxGroups && kWave / S < tn) S >>= 1;
    const int tpw = kWave / S;       // targets per round
    const int tg = lane / S;         // this lane's target slot in the round
    const int sl = lane % S;         // this lane's slice of the tile
    constexpr int U = 4;
    constexpr double kU = 0x1p-24;
    const int nbatch = (total + U * S - 1) / (U * S);
    const int bpl = kHistBuckets / S;  // histogram buckets per lane in the scan (S = 64 -> 1)

    // first round's targets: cell-sorted copies of the coordinates (contiguous, no indirection);
    // issued before the tile loads so that both are in flight together
    double npx, npy, npz, npw;
    {
        const bool v = tg < tn;
        const double2 *r2 = reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + (v ? tg : 0)) * kRec);
        const double2 xy = r2[0], zw = r2[1];
        npx = xy.x;
        npy = xy.y;
        npz = zw.x;
        npw = zw.y;
    }

    // ---- stage the tile: every run holds at most 64 sources, so lane l fetches source l of each
    // run (three runs' loads in flight at a time keeps the register footprint small)
    {
        int off = 0;
#pragma unroll
        for (int c3 = 0; c3 < 9; c3 += 3) {
            double sx[3], sy[3], sz[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const i64 s = (i64)rs[c3 + c] + min(lane, max(rl[c3 + c] - 1, 0));
                const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + s * kRec);
                const double2 xy = r2[0];
                sx[c] = xy.x;
                sy[c] = xy.y;
                sz[c] = r2[1].x;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (lane < rl[c3 + c])
                    tile[off + lane] = make_float4((float)(sx[c] - ox), (float)(sy[c] - oy), (float)(sz[c] - oz),
                                                   __int_as_float(rs[c3 + c] + lane));
                off += rl[c3 + c];
            }
        }
    }
    if (lane == 0) tile[total] = make_float4(1e30f, 1e30f, 1e30f, 0.f);  // slots past the end read this
    if (dbg_stop == 1) return;  // diagnostic builds only (MM_KNN_DBG_STOP): time the phases

    for (int r0 = 0; r0 < tn; r0 += tpw) {
        const int tt = r0 + tg;
        const bool valid = tt < tn;
        const i64 i = valid ? (i64)record_id(npw) : 0;  // the target's original index
        const double px = valid ? npx : ox;
        const double py = valid ? npy : oy;
        const double pz = valid ? npz : oz;
        if (r0 + tpw < tn) {
            // next round's targets, in flight during this round
            const bool v = tt + tpw < tn;
            const double2 *r2 = reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + (v ? tt + tpw : 0)) * kRec);
            const double2 xy = r2[0], zw = r2[1];
            npx = xy.x;
            npy = xy.y;
            npz = zw.x;
            npw = zw.y;
        }
        const float tx = (float)(px - ox), ty = (float)(py - oy), tz = (float)(pz - oz);
        const double E = 3.0 * kU * (fabs(px - ox) + fabs(py - oy) + fabs(pz - oz) + 2.0 * (g.hx + g.hy + g.hz));

        for (int q = lane; q < (kHistBuckets + 1) * kMaxGroups; q += kWave) (&s_hist[0][0])[q] = 0u;
        if (lane < kMaxGroups) {
            s_jb[lane] = kHistBuckets;
            s_seen[lane] = 0ull;
        }
        wave_sync();  // tile staged (first round), counters cleared

        // ---- P1: histogram of fp32 squared distances; the bucket numbers of a lane's slots are
        // kept (4 per word) in LDS for P2.  Slots past the end of the tile read the far-away
        // sentinel, and an idle group's target is moved far away, so the loop has no liveness tests:
        // such pairs fall into the last bucket, which is never counted nor collected.
        const float qx = valid ? tx : 1e30f;
        for (int m = 0; m < nbatch; ++m) {
            float4 q4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) q4[u] = tile[min(sl + (m * U + u) * S, total)];
            unsigned packed = 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float fx = q4[u].x - qx, fy = q4[u].y - ty, fz = q4[u].z - tz;
                const float a = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                // NaN -> last bucket (fminf returns the non-NaN operand).  The last bucket means
                // "beyond the histogram range": most candidates land there, and counting them
                // would serialise the LDS atomic on one address, so they are not counted.
                const int b = (int)fminf(a * scale, (float)(kHistBuckets - 1));
                if (b < kHistBuckets - 1) atomicAdd(&s_hist[b][tg], 1u);
                packed |= (unsigned)b << (8 * u);
            }
            s_pk[m][lane] = packed;
        }
        wave_sync();
        if (dbg_stop == 2 || (dbg_stop >= 20 && dbg_stop <= 23)) return;

        // ---- jb = first bucket whose running count reaches k: each lane sums its share of the
        // buckets, a group prefix sum locates the lane whose share crosses k
        {
            int mine = 0;
            for (int q = 0; q < bpl; ++q) mine += (int)s_hist[sl * bpl + q][tg];
            const int incl = group_scan(mine, sl, S);
            int run_count = incl - mine;
            if (run_count < kout && incl >= kout) {
                for (int q = 0; q < bpl; ++q) {
                    run_count += (int)s_hist[sl * bpl + q][tg];
                    if (run_count >= kout) {
                        s_jb[tg] = sl * bpl + q;
                        break;
                    }
                }
            }
        }
        wave_sync();
        const int jb = s_jb[tg];
        bool hand_over = jb >= kHistBuckets - 2;  // k-th distance beyond the histogram range
        {
            // every exact k-nearest candidate must land in a bucket <= jb+1 (header comment)
            const double e1 = (double)(jb + 1) / (double)scale;
            const double e2 = (double)(jb + 2) / (double)scale;
            const double D = sqrt(e1) * (1.0 + 4.0 * kU) + E;
            const double D2 = D * (1.0 + 4.0 * kU) + E;
            if (!(D2 * D2 * (1.0 + 8.0 * kU) < e2)) hand_over = true;
        }
        if (dbg_stop == 3) { if (jb == 77) fb_list[0] = jb; return; }

        // ---- P2: candidates in buckets <= jb+1 go to the target's list.  Each lane marks its
        // qualifying slots in a bit mask; a group prefix sum of the counts gives the list offsets.
        unsigned qmask = 0u;
        for (int m = 0; m < nbatch; ++m) {
            const unsigned packed = s_pk[m][lane];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = (int)((packed >> (8 * u)) & 0xffu);
                qmask |= (b <= jb + 1 ? 1u : 0u) << (m * U + u);
            }
        }
        if (hand_over) qmask = 0u;
        const int mycnt = __popc(qmask);
        const int incl = group_scan(mycnt, sl, S);
        const int n = __shfl(incl, tg * S + S - 1);
        int pos = incl - mycnt;
        while (qmask) {
            const int slot = __ffs(qmask) - 1;
            qmask &= qmask - 1u;
            if (pos < CAP) s_bx[pos][tg] = __float_as_int(tile[sl + slot * S].w);
            ++pos;
        }
        if (sl == 0) s_cnt[tg] = n;
        wave_sync();
        if (dbg_stop == 4) return;
        if (n > CAP) hand_over = true;
        // widest list in this round (uniform loop bounds below)
        int nmax = 0;
        for (int q = 0; q < tpw; ++q) nmax = max(nmax, min(s_cnt[q], CAP));
        const int owned = (nmax + S - 1) / S;  // list entries per lane: sl, sl+S, ...

        // ---- exact fp64 distance (reference arithmetic) and source id of the owned entries
        constexpr int MAXE = (CAP + 7) / 8;  // owned entries per lane at the narrowest split
        double ed[MAXE];
        int ei[MAXE], rank[MAXE];
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            const bool live = o < owned && e < n && e < CAP;
            ed[o] = INFINITY;
            ei[o] = 0x7fffffff;
            rank[o] = 0;
            if (live) {
                const i64 s = (i64)s_bx[e][tg];
                const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + s * kRec);
                const double2 xy = r2[0], zw = r2[1];
                const double dx = xy.x - px;
                const double dy = xy.y - py;
                const double dz = zw.x - pz;
                double d2 = dx * dx;
                d2 = d2 + dy * dy;
                if (ndim > 2) d2 = d2 + dz * dz;
                ed[o] = d2;
                ei[o] = record_id(zw.y);
                s_bd[e][tg] = d2;
                s_bx[e][tg] = ei[o];
            }
        }
        wave_sync();
        if (dbg_stop == 5) return;

        // ---- P3: rank by exact d2: list entries are read four at a time (broadcast within the
        // group) and compared against the owned ones
        for (int j0 = 0; j0 < nmax; j0 += U) {
            double dj[U];
#pragma unroll
            // (entries at or beyond the capacity do not exist: an overflowing list is handed over, and
            // counting its clamped last entry more than once would push ranks past the row)
            for (int u = 0; u < U; ++u) dj[u] = j0 + u < min(n, CAP) ? s_bd[min(j0 + u, CAP - 1)][tg] : INFINITY;
#pragma unroll
            for (int o = 0; o < MAXE; ++o) {
                if (o < owned) {
#pragma unroll
                    for (int u = 0; u < U; ++u) rank[o] += dj[u] < ed[o] ? 1 : 0;
                }
            }
        }
        if (dbg_stop == 8) return;
        // distinct distances <=> the ranks are a permutation of 0..n-1
        const unsigned long long full = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            if (o < owned && e < n && e < CAP) atomicOr(&s_seen[tg], 1ull << rank[o]);
        }
        wave_sync();
        const bool tied = valid && !hand_over && s_seen[tg] != full;
        if (__any(tied)) {
            // bit-equal distances somewhere in this round: redo the ranks lexicographically
#pragma unroll
            for (int o = 0; o < MAXE; ++o) rank[o] = 0;
            for (int j = 0; j < nmax; ++j) {
                const bool live = j < n;
                const double dj = live ? s_bd[min(j, CAP - 1)][tg] : INFINITY;
                const int ij = live ? s_bx[min(j, CAP - 1)][tg] : 0x7fffffff;
#pragma unroll
                for (int o = 0; o < MAXE; ++o)
                    if (o < owned) rank[o] += before(dj, ij, ed[o], ei[o]) ? 1 : 0;
            }
            wave_sync();
        }
        if (dbg_stop == 9) return;
        // sorted order back into the list (every lane has finished reading it)
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            if (o < owned && e < n && e < CAP) {
                s_bd[rank[o]][tg] = ed[o];
                s_bx[rank[o]][tg] = ei[o];
            }
        }
        wave_sync();
        if (dbg_stop == 10) return;
        if (valid && !hand_over) {
            // the group's lanes write the target's row side by side (coalesced 8-byte stores)
            IDX *row = idx_out + i * kout;
            double *drow = dist_out ? dist_out + i * kout : nullptr;
            if (sizeof(IDX) == 4 && (kout & 3) == 0) {
                // int32 rows (fused pipeline): 16-byte stores of four ids
                for (int e = 4 * sl; e < kout; e += 4 * S)
                    *reinterpret_cast<int4 *>(row + e) =
                        make_int4(s_bx[e][tg], s_bx[e + 1][tg], s_bx[e + 2][tg], s_bx[e + 3][tg]);
            } else if (sizeof(IDX) == 8 && (kout & 1) == 0) {
                // 16-byte stores (rows are 16-byte aligned when k is even): fewer, fuller writes
                for (int e = 2 * sl; e < kout; e += 2 * S) {
                    *reinterpret_cast<longlong2 *>(row + e) = make_longlong2((i64)s_bx[e][tg], (i64)s_bx[e + 1][tg]);
                    if (drow)
                        *reinterpret_cast<double2 *>(drow + e) = make_double2(sqrt(s_bd[e][tg]), sqrt(s_bd[e + 1][tg]));
                }
            } else {
                for (int e = sl; e < kout; e += S) {
                    row[e] = (IDX)s_bx[e][tg];
                    if (drow) drow[e] = sqrt(s_bd[e][tg]);
                }
            }
            if (sizeof(IDX) == 4 && (kout & 3) == 0 && drow)
                for (int e = sl; e < kout; e += S) drow[e] = sqrt(s_bd[e][tg]);
        }
        if (dbg_stop == 11) return;
        if (valid && sl == 0) {
            if (!hand_over) {
                // could a nearer source sit outside the 3x3x3 block?
                const bool all_x = (cx - 1 <= 0) && (cx + 1 >= g.nx - 1);
                const bool all_y = (cy - 1 <= 0) && (cy + 1 >= g.ny - 1);
                const bool all_z = (cz - 1 <= 0) && (cz + 1 >= g.nz - 1);
                if (!(all_x && all_y && all_z)) {
                    const double kth = s_bd[kout - 1][tg];
                    const double bound = block_bound(g, px, py, pz, cx, cy, cz, 1);
                    if (!(bound > 0.0 && kth < bound * bound)) hand_over = true;
                }
            }
            if (hand_over) fb_list[atomicAdd(fb_count, 1)] = (int)i;
        }
        wave_sync();  // before the next round clears the counters
    }
}

// ---- fast path, 3-D grids: the cell kernel's rounds over a STRIP of kStripZ cells along z --------
// The cell kernel pays its fixed costs per cell: a workgroup launch, the metadata loads, staging
// all 27 neighbour cells (of which 18 are shared with the next cell up), and -- for the ~40 % of
// cells holding more than 8 targets -- a second, nearly empty round.  Here one wave owns kStripZ
// consecutive cells of a column:
//   tile   : the strip's cells and their neighbours, (kStripZ+2) layers x 9 columns, staged once and
//            stored LAYER-major, so the 27 cells around a target's cell are one contiguous window
//            [layer(cz-1), layer(cz+2)) of the tile; coordinates relative to the strip's corner.
//   rounds : the strip's targets are taken 8 at a time regardless of their cell (each group walks
//            its own target's window); the last round of a strip widens the split (S = 16..64
//            lanes per target) so that a round for one or two left-over targets is short.
// Everything inside a round (P1 histogram, jb, P2 list, exact fp64, P3 rank sort, error bound) is the
// cell kernel's, see there; only the fp32 rounding bound E uses the strip's extent in z.
#ifndef MM_STRIP_Z          // tuning builds only (make EXTRA="-DMM_STRIP_Z=4 -DMM_STRIP_CAP=496")
#define MM_STRIP_Z 2
#define MM_STRIP_CAP 352   // 36 cells x ~8 expected = 288, + 3 sigma
#endif
constexpr int kStripZ = MM_STRIP_Z;
constexpr int kStripLayers = kStripZ + 2;
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kStripTileCap = MM_STRIP_CAP;
#ifndef MM_STRIP_GROUPS
#define MM_STRIP_GROUPS 8
#endif
constexpr int kStripGroups = MM_STRIP_GROUPS;            // targets per round at the narrowest split
constexpr int kStripSlots = kTileCap / (kWave / kStripGroups);   // window entries per lane at the narrowest split
static_assert(kStripLayers * 9 <= kWave, "one lane stages one tile cell");

// MODE 0: one grid (the common case: no code for anything else).  1: level 0 of a graded cloud (targets
// whose strip is too full are passed down).  2: a denser level -- the workgroups walk the list of the
// strips that hold targets instead of being one workgroup per strip of the (mostly empty) grid.
template <int K, int CAP, typename IDX, int MODE>
__global__ __launch_bounds__(kWave, 4) void knn_strip_kernel(GridParams g, i64 nsrc,
                                                             const int *__restrict__ cell_start,
                                                             const double *__restrict__ sorted_xyz, int ndim,
                                                             int kout, const int *__restrict__ tstart,
                                                             const double *__restrict__ tsorted,
                                                             IDX *__restrict__ idx_out,
                                                             double *__restrict__ dist_out,
                                                             int *__restrict__ fb_list, int *__restrict__ fb_count,
                                                             int dbg_stop, int nsplit, int *__restrict__ down_list,
                                                             int *__restrict__ down_count,
                                                             const unsigned *__restrict__ strip_list,
                                                             const int *__restrict__ strip_count)
{
    static_assert(CAP <= 64, "rank mask is 64 bits");
    constexpr bool WALK = MODE == 2;
    if (MODE == 0) down_list = nullptr;
#ifndef MM_STRIP_NB_SMALL   // tuning builds only
#define MM_STRIP_NB_SMALL 32
#endif
    // histogram buckets: short lists need less resolution (the two buckets collected beyond the k-th
    // distance hold ~1.5 k / buckets * 2.2 candidates each)
    constexpr int kNB = K <= 8 ? MM_STRIP_NB_SMALL : kHistBuckets;
    // log2 of the lanes per target: the widest split (at most one lane per histogram bucket) whose
    // round still covers `left` targets, at least kWave / kStripGroups lanes
    constexpr int kLgMax = kNB >= kWave ? 6 : (kNB >= 32 ? 5 : 4);
    constexpr int kLgMin = kStripGroups == 8 ? 3 : (kStripGroups == 16 ? 2 : 4);
    auto split_log2 = [](int left) {
        int lg = kLgMax;
        while (lg > kLgMin && (kWave >> lg) < left) --lg;
        return lg;
    };
    // The tile holds the sources in PAIRS, {x0,x1,y0,y1}{z0,z1,w0,w1} (w = position in the sorted
    // array), so that P1 evaluates two candidates per packed-fp32 instruction.  Every layer starts
    // at an even entry (an odd layer is padded with one far-away sentinel).  Slots past the end of
    // the tile read the sentinel pair behind it; slots past a window but inside the tile are sources
    // of the next layer -- real candidates, just not needed.
    constexpr int kPairCap = kStripTileCap / 2;
    __shared__ float4 tile_xy[kPairCap + 1];   // separate arrays: consecutive lanes, consecutive words
    __shared__ float2 tile_z[kPairCap + 1];
    __shared__ int2 tile_w[kPairCap + 1];      // only P2 looks at the positions
    float *const txy = reinterpret_cast<float *>(tile_xy);
    float *const tz_ = reinterpret_cast<float *>(tile_z);
    int *const tw_ = reinterpret_cast<int *>(tile_w);
    // Per-target arrays are laid out [group][entry] with strides that spread a group's lanes over
    // the LDS banks (an [entry][group] layout puts the 8 lanes of a group on 2-4 banks).
    // Two pairs of arrays are never live together and share their memory (more waves per CU):
    //   s_pk (P1 -> P2: bucket numbers of each lane's slots)  |  s_bd (exact -> output: distances)
    //   s_hist (P1 -> scan: histogram, last column = sink)    |  s_bx (P2 -> output: positions/ids)
    // Each hand-over is separated by a wave_sync() from the last use of the other member.
    constexpr int kBdStride = CAP | 1;                 // doubles per group (odd)
    constexpr int kBxStride = (CAP + 7) / 4 * 4;       // ints per group (rows stay 16-byte aligned)
    constexpr int kHistStride = kNB + 1;      // words per group (odd)
    constexpr int kPkBytes = (kStripSlots / 4) * kWave * 4, kBdBytes = kStripGroups * kBdStride * 8;
    constexpr int kHistBytes = kStripGroups * kHistStride * 4, kBxBytes = kStripGroups * kBxStride * 4;
    __shared__ __attribute__((aligned(16))) unsigned char s_mem0[kPkBytes > kBdBytes ? kPkBytes : kBdBytes];
    __shared__ __attribute__((aligned(16))) unsigned char s_mem1[kHistBytes > kBxBytes ? kHistBytes : kBxBytes];
    unsigned (*const s_pk)[kWave] = reinterpret_cast<unsigned (*)[kWave]>(s_mem0);
    double (*const s_bd)[kBdStride] = reinterpret_cast<double (*)[kBdStride]>(s_mem0);
    unsigned (*const s_hist)[kHistStride] = reinterpret_cast<unsigned (*)[kHistStride]>(s_mem1);
    int (*const s_bx)[kBxStride] = reinterpret_cast<int (*)[kBxStride]>(s_mem1);
    __shared__ int s_jb[kStripGroups];
    __shared__ int s_cnt[kStripGroups];
    __shared__ unsigned long long s_seen[kStripGroups];
    __shared__ int s_layer[kStripLayers + 1];

    const int lane = threadIdx.x;
    if (dbg_stop == 100) return;   // diagnostic: what dispatching the grid alone costs
    // XCD-aware strip -> workgroup map (see knn_cell_kernel): XCD x owns a slab of columns
    const int ncols = g.nx * g.ny;
    const int cols_per_xcd = (ncols + 7) / 8;
    const int nstrips = (g.nz + kStripZ - 1) / kStripZ;
    // With strip_list (the denser levels of a graded cloud, whose grids are mostly empty) the workgroups
    // walk the list of strips that hold targets -- entries in the encoding of blockIdx.x -- instead of
    // being one workgroup per strip of the grid.
    for (unsigned sidx = blockIdx.x;; sidx += gridDim.x) {
    if (WALK) {
        if (sidx >= (unsigned)*strip_count) break;
        if (sidx != blockIdx.x) wave_sync();   // the previous strip's LDS is done with
    }
#define MM_NEXT_STRIP { if (!WALK) return; continue; }
    const unsigned bid = WALK ? strip_list[sidx] : blockIdx.x;
    const int xcd = bid & 7;
    // nsplit > 1 (many more targets than sources, e.g. the unique GLL points of a fine mesh over a
    // coarse one): nsplit waves per strip -- consecutive workgroups of one XCD -- share its targets
    int m = bid >> 3, part = 0;
    if (nsplit > 1) {
        part = m % nsplit;
        m = m / nsplit;
    }
    const int colm = m / nstrips;
    const int col = xcd * cols_per_xcd + colm;
    if (colm >= cols_per_xcd || col >= ncols) MM_NEXT_STRIP
    const int strip = m - colm * nstrips;
    const int cx = col / g.ny, cy = col - cx * g.ny;
    const int cz0 = strip * kStripZ, cz1 = min(cz0 + kStripZ, g.nz);
    // the tile's cell extents are requested before the strip's target range is looked at: both
    // round trips are in flight together (a strip without targets throws them away)
    const int za = max(cz0 - 1, 0), zb = min(cz1, g.nz - 1);
    const int nlayers = zb - za + 1;
    const int ntc = nlayers * 9;
    const int layer = lane / 9, c = lane - layer * 9;
    int s0 = 0, cnt = 0;
    {
        const int ix = cx + c / 3 - 1, iy = cy + (c - (c / 3) * 3) - 1;
        const bool inside = lane < ntc && (unsigned)ix < (unsigned)g.nx && (unsigned)iy < (unsigned)g.ny;
        if (inside) {
            const int cellid = (ix * g.ny + iy) * g.nz + za + layer;
            s0 = cell_start[cellid];
            cnt = cell_start[cellid + 1] - s0;
        }
    }
    int t0 = tstart[col * g.nz + cz0];
    const int t1 = tstart[col * g.nz + cz1];
    int tn = t1 - t0;
    if (nsplit > 1) {
        // whole rounds of kStripGroups targets per part
        const int chunk = ((tn + nsplit - 1) / nsplit + kStripGroups - 1) / kStripGroups * kStripGroups;
        t0 += part * chunk;
        tn = min(chunk, t1 - t0);
    }
    if (tn <= 0) MM_NEXT_STRIP
    const double ox = g.lox + (double)cx * g.hx;
    const double oy = g.loy + (double)cy * g.hy;
    const double oz = g.loz + (double)cz0 * g.hz;

    // first round's targets: cell-sorted copies of the coordinates (contiguous, no indirection);
    // issued before the tile loads so that both are in flight together
    double npx, npy, npz, npw;
    {
        const int tg1 = lane >> split_log2(tn);
        const double2 *r2 = reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + (tg1 < tn ? tg1 : 0)) * kRec);
        const double2 xy = r2[0], zw = r2[1];
        npx = xy.x;
        npy = xy.y;
        npz = zw.x;
        npw = zw.y;
    }

    // ---- stage the tile: lane l copies cell l of the (layer, column) list
    int total;
    {
        int incl = cnt;
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        // pad odd layers: entries of layer L shift by the number of odd layers below it
        int pad = 0, pads_all = 0;
        for (int L = 0; L < nlayers; ++L) {
            const int end = __shfl(incl, 9 * L + 8);
            const int beg = L > 0 ? __shfl(incl, 9 * L - 1) : 0;
            const int odd = (end - beg) & 1;
            if (L < layer) pad += odd;
            pads_all += odd;
        }
        total = __shfl(incl, kWave - 1) + pads_all;
        const int off = incl - cnt + pad;
        if (lane < ntc && c == 0) s_layer[layer] = off;
        if (lane == 0) s_layer[nlayers] = total;
        if (total <= kStripTileCap) {
            // eight records per trip: a cell holds ~8 sources, so most strips need a single trip and
            // all of its loads are in flight together (the registers are free before the rounds start)
            constexpr int kCopy = 8;
            for (int q = 0; __any(q < cnt); q += kCopy) {
                double2 xy[kCopy], zw[kCopy];
#pragma unroll
                for (int u = 0; u < kCopy; ++u) {
                    const i64 s = (i64)s0 + min(q + u, max(cnt - 1, 0));
                    const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + s * kRec);
                    xy[u] = r2[0];
                    zw[u] = r2[1];
                }
#pragma unroll
                for (int u = 0; u < kCopy; ++u)
                    if (q + u < cnt) {
                        const int e = off + q + u;
                        const int at = (e >> 1) * 4 + (e & 1);
                        txy[at] = (float)(xy[u].x - ox);
                        txy[at + 2] = (float)(xy[u].y - oy);
                        tz_[e] = (float)(zw[u].x - oz);
                        tw_[e] = s0 + q + u;
                    }
            }
            // the sentinel that evens out an odd layer (written by the layer's last cell)
            if (lane < ntc && c == 8 && ((off + cnt) & 1)) {
                const int at = ((off + cnt) >> 1) * 4 + 1;
                txy[at] = 1e30f;
                txy[at + 2] = 1e30f;
                tz_[off + cnt] = 1e30f;
                tw_[off + cnt] = 0;
            }
            if (lane == 0) {
                tile_xy[total >> 1] = make_float4(1e30f, 1e30f, 1e30f, 1e30f);
                tile_z[total >> 1] = make_float2(1e30f, 1e30f);
                tile_w[total >> 1] = make_int2(0, 0);
            }
        }
    }
    if (total > kStripTileCap || total < kout) {
        // the whole strip goes to the next density level when it is too full for the tile and there is
        // one, else to the generic kernel
        // (the choice is made wave-uniform explicitly and each branch names its counter directly:
        // `total` comes out of shuffles, and with a selected pointer the compiler does not combine the
        // lanes' atomics into one per wave -- 21 ms of same-address atomics on a graded cloud)
        const bool down = __builtin_amdgcn_readfirstlane((int)(total > kStripTileCap)) != 0 && down_list != nullptr;
        if (down) {
            for (int q = lane; q < tn; q += kWave)
                down_list[atomicAdd(down_count, 1)] = record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
        } else {
            for (int q = lane; q < tn; q += kWave)
                fb_list[atomicAdd(fb_count, 1)] = record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
        }
        MM_NEXT_STRIP
    }
    wave_sync();  // tile and layer table staged
    if (dbg_stop == 1) return;  // diagnostic builds only (MM_KNN_DBG_STOP): time the phases

    // widest window of the strip's cells (uniform loop bounds), block volume per layer
    int maxwin = 0;
    for (int cz = cz0; cz < cz1; ++cz)
        maxwin = max(maxwin, s_layer[min(cz + 1, zb) - za + 1] - s_layer[max(cz - 1, za) - za]);
    maxwin = min(maxwin, kTileCap);
    int dims = 0;
    float vol_layer = 1.f;
    {
        const int bx = min(cx + 1, g.nx - 1) - max(cx - 1, 0) + 1;
        const int by = min(cy + 1, g.ny - 1) - max(cy - 1, 0) + 1;
        if (g.nx > 1) { ++dims; vol_layer *= (float)bx * (float)g.hx; }
        if (g.ny > 1) { ++dims; vol_layer *= (float)by * (float)g.hy; }
        if (g.nz > 1) { ++dims; vol_layer *= (float)g.hz; }
    }
    constexpr int U = 4;
    static_assert(U == 4, "nbatch uses a shift by log2(U)");
    constexpr double kU = 0x1p-24;

    int tpw = 0;
    for (int r0 = 0; r0 < tn; r0 += tpw) {
        // lanes per target: the widest split whose round still covers the remaining targets
        const int rem = tn - r0;
        // (S is a power of two: shifts, not the integer divisions a runtime S would cost)
        const int lgS = split_log2(rem);
        const int S = 1 << lgS;
        tpw = kWave >> lgS;
        const int tg = lane >> lgS;      // this lane's target slot in the round
        const int sl = lane & (S - 1);   // this lane's slice of the window
        const int nbatch = (maxwin + U * S - 1) >> (lgS + 2);   // U = 4
        const int bpl = kNB >> lgS;      // histogram buckets per lane in the scan
        const bool valid = tg < rem;
        const double px = valid ? npx : ox;
        const double py = valid ? npy : oy;
        const double pz = valid ? npz : oz;
        const i64 i = valid ? (i64)record_id(npw) : 0;  // the target's original index
        if (rem > tpw) {
            // next round's targets (its split may be wider), in flight during this round
            const int rem2 = rem - tpw;
            const int tg2 = lane >> split_log2(rem2);
            const double2 *r2 =
                reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + r0 + tpw + (tg2 < rem2 ? tg2 : 0)) * kRec);
            const double2 xy = r2[0], zw = r2[1];
            npx = xy.x;
            npy = xy.y;
            npz = zw.x;
            npw = zw.y;
        }
        for (int q = lane; q < (kNB + 1) * kStripGroups; q += kWave) (&s_hist[0][0])[q] = 0u;
        if (lane < kStripGroups) {
            s_jb[lane] = kNB;
            s_seen[lane] = 0ull;
        }
        const int czl = min(max(cell_coord(pz, g.loz, g.ihz, g.nz), cz0), cz1 - 1);
        const int l0 = max(czl - 1, za) - za, l1 = min(czl + 1, zb) - za + 1;
        const int ws = s_layer[l0];
        const int we = valid ? s_layer[l1] : ws;
        // an idle group's target is moved far away: all its pairs fall into the (uncounted) last bucket
        const float tx = valid ? (float)(px - ox) : -1e30f, ty = (float)(py - oy), tz = (float)(pz - oz);
        const double E = 3.0 * kU * (fabs(px - ox) + fabs(py - oy) + fabs(pz - oz) + 2.0 * (g.hx + g.hy) +
                                     (double)(kStripZ + 1) * g.hz);
        // histogram range from the density of the target's own window: the ball holding k of the
        // window's sources has r^d = (k/count) * V_block / c_d; buckets are uniform in r^2 over
        // [0, 2.2 r^2).  Only a heuristic range, so fast exp2/log2 are fine.
        float scale, width;
        {
            const float vol = g.nz > 1 ? vol_layer * (float)(l1 - l0) : vol_layer;
            const float frac = (float)kout / (float)max(we - ws, 1);
            float r2;
            if (dims == 3) r2 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(frac * vol * (1.f / 4.18879f)) * (2.f / 3.f));
            else if (dims == 2) r2 = frac * vol * (1.f / 3.14159f);
            else if (dims == 1) { const float r = frac * vol * 0.5f; r2 = r * r; }
            else r2 = 1.f;
            // bucket width first (the division by the power-of-two bucket count is exact), the binning
            // factor is its correctly rounded reciprocal: 1/scale = width (1 +- 2u)
            width = 2.2f * r2 * (1.f / (float)kNB);
            scale = 1.f / width;
        }
        const bool too_dense = we - ws > kTileCap;   // the window alone is more than a round can take
        bool hand_over = !(scale > 0.f && scale < INFINITY) || we - ws < kout || too_dense;
        if (!(scale > 0.f && scale < INFINITY)) scale = width = 1.f;
        wave_sync();  // counters cleared

        // ---- P1: histogram of fp32 squared distances (two candidates per packed instruction); the
        // bucket numbers of a lane's slots are kept (4 per word) in LDS for P2.  Slot 4m+u of a lane
        // is half (u & 1) of pair wsp + sl + (2m + u/2) * S.
        const int wsp = ws >> 1, total_p = total >> 1;
        const v2f tx2 = {tx, tx}, ty2 = {ty, ty}, tz2 = {tz, tz}, scale2 = {scale, scale};
        for (int m = 0; m < nbatch; ++m) {
            float4 qa[2];
            float2 qb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pr = min(wsp + sl + (2 * m + h) * S, total_p);
                qa[h] = tile_xy[pr];
                qb[h] = tile_z[pr];
            }
            unsigned packed = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2f fx = v2f{qa[h].x, qa[h].y} - tx2;
                const v2f fy = v2f{qa[h].z, qa[h].w} - ty2;
                const v2f fz = v2f{qb[h].x, qb[h].y} - tz2;
                const v2f a = __builtin_elementwise_fma(fz, fz, __builtin_elementwise_fma(fy, fy, fx * fx));
                const v2f sc = a * scale2;
                // NaN -> last bucket (fminf returns the non-NaN operand).  The last bucket means
                // "beyond the histogram range": most candidates land there, and counting them
                // would serialise the LDS atomic on one address, so they are not counted.
                const int b0 = (int)fminf(sc.x, (float)(kNB - 1));
                const int b1 = (int)fminf(sc.y, (float)(kNB - 1));
                if (b0 < kNB - 1) atomicAdd(&s_hist[tg][b0], 1u);
                if (b1 < kNB - 1) atomicAdd(&s_hist[tg][b1], 1u);
                packed |= ((unsigned)b0 | ((unsigned)b1 << 8)) << (16 * h);
            }
            s_pk[m][lane] = packed;
        }
        wave_sync();
        if (dbg_stop == 2) return;

        // ---- jb = first bucket whose running count reaches k
        {
            int mine = 0;
            for (int q = 0; q < bpl; ++q) mine += (int)s_hist[tg][sl * bpl + q];
            const int incl = group_scan(mine, sl, S);
            int run_count = incl - mine;
            if (run_count < kout && incl >= kout) {
                for (int q = 0; q < bpl; ++q) {
                    run_count += (int)s_hist[tg][sl * bpl + q];
                    if (run_count >= kout) {
                        s_jb[tg] = sl * bpl + q;
                        break;
                    }
                }
            }
        }
        wave_sync();
        const int jb = s_jb[tg];
        if (jb >= kNB - 2) hand_over = true;  // k-th distance beyond the histogram range
        {
            // every exact k-nearest candidate must land in a bucket <= jb+1 (cell kernel's header).  No
            // fp64 division or square root here (once per target and round, they were a tenth of the
            // round's instructions): e1 >= (jb+1)/scale and e2 <= (jb+2)/scale from the bucket width,
            // and an fp32 square root rounded up bounds sqrt(e1) from above.
            const double e1 = (double)(jb + 1) * (double)width * (1.0 + 4.0 * kU);
            const double e2 = (double)(jb + 2) * (double)width * (1.0 - 4.0 * kU);
            const double root = (double)__builtin_sqrtf((float)(e1 * (1.0 + 2.0 * kU))) * (1.0 + 4.0 * kU);
            const double D = root * (1.0 + 4.0 * kU) + E;
            const double D2 = D * (1.0 + 4.0 * kU) + E;
            if (!(D2 * D2 * (1.0 + 8.0 * kU) < e2)) hand_over = true;
        }
        if (dbg_stop == 3) { if (jb == 77) fb_list[0] = jb; return; }

        // ---- P2: candidates in buckets <= jb+1 go to the target's list
        // (four bucket numbers per word, each < 64: adding 126 - jb sets a byte's top bit exactly when
        // its bucket is >= jb + 2, without carries; the multiply gathers the four flags)
        unsigned long long qmask = 0ull;
        {
            const unsigned bias = (unsigned)(126 - min(jb, kNB)) * 0x01010101u;
            for (int m = 0; m < nbatch; ++m) {
                const unsigned keep = (~(s_pk[m][lane] + bias) & 0x80808080u) >> 7;
                qmask |= (unsigned long long)(((keep * 0x00204081u) >> 21) & 0xfu) << (m * U);
            }
        }
        if (hand_over) qmask = 0ull;
        const int mycnt = __popcll(qmask);
        const int incl = group_scan(mycnt, sl, S);
        const int n = __shfl(incl, tg * S + S - 1);
        int pos = incl - mycnt;
        while (qmask) {
            const int slot = __ffsll((long long)qmask) - 1;
            qmask &= qmask - 1ull;
            if (pos < CAP)
                s_bx[tg][pos] = tw_[(wsp + sl + (slot >> 1) * S) * 2 + (slot & 1)];
            ++pos;
        }
        if (sl == 0) s_cnt[tg] = n;
        wave_sync();
        if (dbg_stop == 4) return;
        if (n > CAP) hand_over = true;
        int nmax = 0;
        for (int q = 0; q < tpw; ++q) nmax = max(nmax, min(s_cnt[q], CAP));
        const int owned = (nmax + S - 1) >> lgS;  // list entries per lane: sl, sl+S, ...

        // ---- exact fp64 distance (reference arithmetic) and source id of the owned entries
        constexpr int MAXE = (CAP + kWave / kStripGroups - 1) / (kWave / kStripGroups);  // at the narrowest split
        double ed[MAXE];
        int ei[MAXE], rank[MAXE];
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            const bool live = o < owned && e < n && e < CAP;
            ed[o] = INFINITY;
            ei[o] = 0x7fffffff;
            rank[o] = 0;
            if (live) {
                const i64 s = (i64)s_bx[tg][e];
                const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + s * kRec);
                const double2 xy = r2[0], zw = r2[1];
                const double dx = xy.x - px;
                const double dy = xy.y - py;
                const double dz = zw.x - pz;
                double d2 = dx * dx;
                d2 = d2 + dy * dy;
                if (ndim > 2) d2 = d2 + dz * dz;
                ed[o] = d2;
                ei[o] = record_id(zw.y);
                s_bd[tg][e] = d2;
                s_bx[tg][e] = ei[o];
            }
        }
        wave_sync();
        if (dbg_stop == 5) return;

        // ---- P3: rank by exact d2
        for (int j0 = 0; j0 < nmax; j0 += U) {
            double dj[U];
#pragma unroll
            // (entries at or beyond the capacity do not exist: an overflowing list is handed over, and
            // counting its clamped last entry more than once would push ranks past the row)
            for (int u = 0; u < U; ++u) dj[u] = j0 + u < min(n, CAP) ? s_bd[tg][min(j0 + u, CAP - 1)] : INFINITY;
#pragma unroll
            for (int o = 0; o < MAXE; ++o) {
                if (o < owned) {
#pragma unroll
                    for (int u = 0; u < U; ++u) rank[o] += dj[u] < ed[o] ? 1 : 0;
                }
            }
        }
        // distinct distances <=> the ranks are a permutation of 0..n-1
        const unsigned long long full = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            if (o < owned && e < n && e < CAP) atomicOr(&s_seen[tg], 1ull << rank[o]);
        }
        wave_sync();
        const bool tied = valid && !hand_over && s_seen[tg] != full;
        if (__any(tied)) {
            // bit-equal distances somewhere in this round: redo the ranks lexicographically
#pragma unroll
            for (int o = 0; o < MAXE; ++o) rank[o] = 0;
            for (int j = 0; j < nmax; ++j) {
                const bool live = j < n;
                const double dj = live ? s_bd[tg][min(j, CAP - 1)] : INFINITY;
                const int ij = live ? s_bx[tg][min(j, CAP - 1)] : 0x7fffffff;
#pragma unroll
                for (int o = 0; o < MAXE; ++o)
                    if (o < owned) rank[o] += before(dj, ij, ed[o], ei[o]) ? 1 : 0;
            }
            wave_sync();
        }
        if (dbg_stop == 6) return;
        // sorted order back into the list (every lane has finished reading it)
#pragma unroll
        for (int o = 0; o < MAXE; ++o) {
            const int e = sl + o * S;
            if (o < owned && e < n && e < CAP) {
                s_bd[tg][rank[o]] = ed[o];
                s_bx[tg][rank[o]] = ei[o];
            }
        }
        wave_sync();
        if (valid && !hand_over) {
            // the group's lanes write the target's row side by side
            IDX *row = idx_out + i * kout;
            double *drow = dist_out ? dist_out + i * kout : nullptr;
            if (sizeof(IDX) == 4 && (kout & 3) == 0) {
                for (int e = 4 * sl; e < kout; e += 4 * S)
                    *reinterpret_cast<int4 *>(row + e) = *reinterpret_cast<const int4 *>(&s_bx[tg][e]);
            } else if (sizeof(IDX) == 8 && (kout & 1) == 0) {
                for (int e = 2 * sl; e < kout; e += 2 * S) {
                    *reinterpret_cast<longlong2 *>(row + e) = make_longlong2((i64)s_bx[tg][e], (i64)s_bx[tg][e + 1]);
                    if (drow)
                        *reinterpret_cast<double2 *>(drow + e) = make_double2(sqrt(s_bd[tg][e]), sqrt(s_bd[tg][e + 1]));
                }
            } else {
                for (int e = sl; e < kout; e += S) {
                    row[e] = (IDX)s_bx[tg][e];
                    if (drow) drow[e] = sqrt(s_bd[tg][e]);
                }
            }
            if (sizeof(IDX) == 4 && (kout & 3) == 0 && drow)
                for (int e = sl; e < kout; e += S) drow[e] = sqrt(s_bd[tg][e]);
        }
        if (dbg_stop == 7) return;
        if (valid && sl == 0) {
            if (!hand_over) {
                // could a nearer source sit outside the target's 3x3x3 block?
                const bool all_x = (cx - 1 <= 0) && (cx + 1 >= g.nx - 1);
                const bool all_y = (cy - 1 <= 0) && (cy + 1 >= g.ny - 1);
                const bool all_z = (czl - 1 <= 0) && (czl + 1 >= g.nz - 1);
                if (!(all_x && all_y && all_z)) {
                    const double kth = s_bd[tg][kout - 1];
                    const double bound = block_bound(g, px, py, pz, cx, cy, czl, 1);
                    if (!(bound > 0.0 && kth < bound * bound)) hand_over = true;
                }
            }
            if (MODE == 0 && hand_over) fb_list[atomicAdd(fb_count, 1)] = (int)i;
        }
        if (MODE != 0) {
            // hand-overs of this round: one atomic per list and wave (with two lists to choose from
            // the compiler no longer combines the lanes' atomics itself)
            const bool push = valid && sl == 0 && hand_over;
            const bool push_down = push && too_dense && down_list != nullptr;
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long md = __ballot(push_down), mf = __ballot(push && !push_down);
            if (md) {
                const int first = __ffsll((long long)md) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(down_count, __popcll(md));
                base = __shfl(base, first);
                if (push_down) down_list[base + __popcll(md & lt)] = (int)i;
            }
            if (mf) {
                const int first = __ffsll((long long)mf) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(fb_count, __popcll(mf));
                base = __shfl(base, first);
                if (push && !push_down) fb_list[base + __popcll(mf & lt)] = (int)i;
            }
        }
        wave_sync();  // before the next round clears the counters
        if (dbg_stop == 8) return;
    }
    if (!WALK) break;
    }
#undef MM_NEXT_STRIP
}

// ---- fast path, 3-D grids, round 2: ONE LANE PER TARGET over an LDS tile ---------------------------
// The strip kernel above spends three quarters of its instructions outside the distance evaluations:
// histogram scans, prefix sums inside lane groups, seven hand-over points per round of 8 targets, a
// dependent global round trip per round.  Here a wave takes 64 targets of a strip of Z cells along z and
// every lane owns ONE target from start to finish -- no cross-lane step inside a round at all:
//   tile   : the strip's cells and their neighbours, (Z+2) layers x 9 columns, staged once per work item
//            as float4 {x, y, z relative to the strip corner, position in the sorted array}, layer-major
//            (the 27 cells around a target's cell are ONE contiguous window, as in the strip kernel).
//   scan   : the lane walks its window (same trip count for the whole wave; a shorter window starts
//            earlier and reads sources of the layer below -- real candidates, just not needed) and keeps
//            the L = K + 2 smallest KEYS in registers, sorted, by one v_med3_f32 per list slot:
//            inserting c into an ascending list is  d[s] = med3(d[s-1], c, d[s]).  A key is the fp32
//            squared distance with its 10 low mantissa bits replaced by the candidate's slot in the
//            window, so the payload rides along for free: 6 + 1 + L VALU per candidate, no LDS write,
//            no atomics, no second pass.
//   exact  : the K + 1 best keys' candidates get the exact fp64 distance in the reference's arithmetic
//            (coordinates re-read from the fp64 records) and are ranked by (d2, id) in registers.
//   certify: every candidate outside the list has a key above the list's last one, B.  With the rounding
//            bound E of the strip kernel (|sqrt(d32) - d| <= E + 2u d) and the 2^-13 the payload can
//            move a key, such a candidate lies at an exact distance >= LB = (sqrt(B)(1 - 2^-12) - E)(1 - 4u).
//            The row is accepted only if the exact k-th distance is strictly below LB (and below the
//            nearest face of the 3x3x3 block, as before); then the list holds every source that can
//            be among the k nearest, exact ties included.  Otherwise (~never on meshes; near-equal
//            k-th .. (k+2)-th distances, hull targets) the target goes to the generic kernel.
// Work items: a prepass turns the strips that hold targets into a list of (strip, part) items of at most
// kLaneRounds rounds each -- so a slab of densely packed targets over 1/8 of the grid (a cfg4 shard) keeps
// the whole chip busy --, and XCD x takes the x-th eighth of the list (contiguous in space: its L2 sees
// each source ~once).
constexpr int kLaneTileCap = 768;      // sources per tile: (7 + 2) layers x 9 columns x ~8 = 648, + 4.7 sigma (Poisson)
constexpr int kLaneTrips = kLaneTileCap / 64;   // staging trips: every lane holds its share of the WHOLE tile in registers
constexpr int kLaneThin = 6;           // thin layers per cell layer (the tile is ordered by them, see the kernel's header)
constexpr int kLaneWin = 5;            // half-width of a target's window in thin layers, first attempt (5/6 of a cell edge: W = 4 is 4 % faster on
                                       // mesh nodes, whose 8 nearest centroids are their own elements', and 25 % slower on random clouds); widened to kLaneThin on demand
constexpr int kLaneThinMax = 64;       // thin layers per tile: one lane each in the prefix sum
constexpr int kLaneUnroll = 8;
constexpr int kLanePad = 16;           // far-away entries behind the tile (a window read may run past it by < 12 entries)
constexpr int kLaneZ = 7;              // cells per strip: ~57 targets per round of 64 lanes at 8 targets per cell
constexpr int kLaneZMax = 12;
constexpr int kLaneRounds = 4;         // rounds (of 64 targets) per work item
constexpr i64 kLaneProbeMin = 32768;              // queries at least this large whose targets are sparse ON AVERAGE are looked at more closely:
constexpr i64 kLaneProbeTargetsPerItem = 16;      // ... the lane kernel serves them when a work item holds at least this many targets
// (the packed prefix sums of the cell counts give each half 16 bits: counts are clamped to kLaneTileCap + 1, the lower
// word sums 64 of them, the upper one the rest of the (Z + 2) x 9 cells)
static_assert(64 * (kLaneTileCap + 1) < 65536 && ((kLaneZMax + 2) * 9 - 64) * (kLaneTileCap + 1) < 65536,
              "knn_lane_kernel: a packed prefix sum of cell counts could wrap");
constexpr float kLaneFar = 1e18f;      // sentinel coordinate (squares to 1e36 < FLT_MAX: keys stay finite)
constexpr float kLaneFarKey = 1e30f;   // keys at or above this are sentinels / absurdly far sources


// Diagnostic builds only (make EXTRA=-DMM_LANE_STAMPS): where a wave of knn_lane_kernel spends its cycles.
// Phase sums (s_memtime ticks = shader cycles) per workgroup slot; tools/lane_stamps.py prints the shares.
#ifdef MM_LANE_STAMPS
constexpr int kStampSlots = 1 << 18;
__device__ unsigned long long g_lane_stamps[kStampSlots * 8];   // per workgroup: 7 phase sums + a wave count
#define MM_STAMP(n)                                                                    \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        stamp_sum[n] += now_ - stamp_last;                                             \
        stamp_last = now_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#else
#define MM_STAMP(n) do { } while (0)
#endif
// Wave-wide inclusive prefix sum / maximum with DPP moves (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then the
// row broadcasts 15 and 31): six VALU instructions with a few cycles of latency each, where __shfl_up is a
// ds_bpermute through the LDS crossbar (~100 cycles each, six of them dependent).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_from(int v, int fill)
{
    // lanes without a source lane (shifted in from outside the row / rows not in ROW_MASK) read `fill`
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v += dpp_from<0x111, 0xF>(v, 0);   // row_shr:1
    v += dpp_from<0x112, 0xF>(v, 0);   // row_shr:2
    v += dpp_from<0x114, 0xF>(v, 0);   // row_shr:4
    v += dpp_from<0x118, 0xF>(v, 0);   // row_shr:8
    v += dpp_from<0x142, 0xA>(v, 0);   // row_bcast:15 -> rows 1 and 3
    v += dpp_from<0x143, 0xC>(v, 0);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// maximum over the wave of non-negative values (every lane's result is only meaningful in lane 63: read it there)
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    v = max(v, dpp_from<0x111, 0xF>(v, 0));
    v = max(v, dpp_from<0x112, 0xF>(v, 0));
    v = max(v, dpp_from<0x114, 0xF>(v, 0));
    v = max(v, dpp_from<0x118, 0xF>(v, 0));
    v = max(v, dpp_from<0x142, 0xA>(v, 0));
    v = max(v, dpp_from<0x143, 0xC>(v, 0));
    return __builtin_amdgcn_readlane(v, 63);
}

// Item q of the list (strips in spatial order) is stored at slot 8 m + x, x = the eighth of the list it lies in,
// m = its place inside that eighth: workgroup b of the lane kernel simply takes slot b -- workgroups are dealt
// round-robin over the 8 XCDs, so XCD x walks the x-th eighth of the list, a contiguous piece of space (its L2
// sees each source about once), and the workgroup's first load depends on nothing but its own index.  Slots
// without an item stay at -1 (the array is pre-set).
// The list is made in THREE dispatches (it used to take six: count, three scan kernels, a fill of the slots, the fill of the
// items): the per-strip item counts are recomputed from the targets' cell starts wherever they are needed.
__device__ __forceinline__ int lane_strip_parts(const GridParams &g, const int *__restrict__ tstart, int Z, int per_item,
                                                i64 t, i64 nstrips_total)
{
    if (t >= nstrips_total) return 0;
    const int nstrips = (g.nz + Z - 1) / Z;
    const int col = (int)(t / nstrips), strip = (int)(t - (i64)col * nstrips);
    const int cz0 = strip * Z, cz1 = min(cz0 + Z, g.nz);
    const int tn = tstart[col * g.nz + cz1] - tstart[col * g.nz + cz0];
    return (tn + per_item - 1) / per_item;
}

// (1) per tile of kScanTile strips: the number of items; every slot of the list is pre-set to "no item" on the way
__global__ __launch_bounds__(kBlock) void lane_items_sums_kernel(GridParams g, const int *__restrict__ tstart, int Z,
                                                                 int per_item, i64 nstrips_total, int *__restrict__ tile_sums,
                                                                 int2 *__restrict__ items, i64 nslots)
{
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int sum = 0;
    for (int q = 0; q < kScanItems; ++q) sum += lane_strip_parts(g, tstart, Z, per_item, base + q, nstrips_total);
    int total;
    (void)block_exclusive_scan(sum, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
    for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < nslots; q += (i64)gridDim.x * blockDim.x)
        items[q] = make_int2(-1, -1);
}

// (2) single block: exclusive scan of the tile sums, the grand total behind them
__global__ __launch_bounds__(kBlock) void lane_items_offsets_kernel(int *__restrict__ tile_sums, int ntiles)
{
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
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

// (3) every strip's items into their slots (the slot rule above)
__global__ __launch_bounds__(kBlock) void lane_items_place_kernel(GridParams g, const int *__restrict__ tstart, int Z,
                                                                  int per_item, i64 nstrips_total,
                                                                  const int *__restrict__ tile_sums, int ntiles,
                                                                  int2 *__restrict__ items)
{
    const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    int np[kScanItems];
    int sum = 0;
    for (int q = 0; q < kScanItems; ++q) {
        np[q] = lane_strip_parts(g, tstart, Z, per_item, base + q, nstrips_total);
        sum += np[q];
    }
    int block_total;
    int a = block_exclusive_scan(sum, &block_total) + tile_sums[blockIdx.x];
    const i64 total = tile_sums[ntiles];
    for (int q = 0; q < kScanItems; ++q) {
        for (int part = 0; part < np[q]; ++part) {
            const i64 it = (i64)a + part;
            int x = (int)((it * 8) / total);
            while (x > 0 && it < ((total * x) >> 3)) --x;
            while (x < 7 && it >= ((total * (x + 1)) >> 3)) ++x;
            const i64 m = it - ((total * x) >> 3);
            items[8 * m + x] = make_int2((int)(base + q), part);
        }
        a += np[q];
    }
}

// neg_inf: -inf in a register the compiler cannot see through -- med3(-inf, c, d0) = min(c, d0) as ONE
// v_med3_f32 (a literal -inf is folded into fminf, which costs two canonicalising v_max_f32 more)
template <int L>
__device__ __forceinline__ void lane_list_insert(float (&d)[L], float c, float neg_inf)
{
#pragma unroll
    for (int s = L - 1; s >= 1; --s) d[s] = __builtin_amdgcn_fmed3f(d[s - 1], c, d[s]);
    d[0] = __builtin_amdgcn_fmed3f(neg_inf, c, d[0]);
}

template <int K, typename IDX>
__global__ __launch_bounds__(kWave, 3) void knn_lane_kernel(GridParams g, i64 nsrc, const int *__restrict__ cell_start,
                                                            const double *__restrict__ sorted_xyz, int ndim, int kout,
                                                            const int *__restrict__ tstart,
                                                            const double *__restrict__ tsorted, IDX *__restrict__ idx_out,
                                                            double *__restrict__ dist_out, int *__restrict__ fb_list,
                                                            int *__restrict__ fb_count, const int2 *__restrict__ items,
                                                            int nslots, int Z, int per_item, int sorted_rows,
                                                            int *__restrict__ down_list, int *__restrict__ down_count,
                                                            int T, int W)
{
    // sorted_rows: a target's row goes to its position in the cell-sorted order (the fused pipeline's locate
    // stage then walks the targets in that order: rows and coordinates stream, neighbours share elements) and
    // hand-overs are queued by that position; otherwise to the target's own index.
    constexpr int L = K + 2;        // keys kept per target
    constexpr int NE = K + 1;       // of which the first K + 1 get exact distances
    static_assert(K >= 1 && NE <= 32, "rank masks are 32 bits");
    constexpr bool kRowsInLds = K > 8;   // short rows are put in rank order in registers (no LDS: one more wave per SIMD)
    constexpr bool kRetry = K <= 8;      // narrow windows first (the launcher passes W = T for the long lists)
    constexpr double kU = 0x1p-24;
    __shared__ float4 tile[kLaneTileCap + kLanePad];   // {x, y, z, position in the sorted array (bits; -1: padding)}
    __shared__ int s_hist[kLaneThinMax];       // entries per thin layer (ranks are handed out by the atomic)
    __shared__ int s_thin[kLaneThinMax + 1];   // first entry of every thin layer; [NL ...] = the tile's length
    __shared__ int s_row[kRowsInLds ? kWave : 1][K | 1];   // long rows in rank order (odd stride: lanes on distinct banks)

    const int lane = threadIdx.x;
#ifdef MM_LANE_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    const int nstrips = (g.nz + Z - 1) / Z;
    const float4 far_entry = make_float4(kLaneFar, kLaneFar, kLaneFar, __int_as_float(-1));
    {
        // ONE work item per workgroup (the grid is the slot count): no loop around the item, so nothing the early
        // phases need -- pointers, extents -- has to stay in scalar registers for a next trip
        if ((int)blockIdx.x >= nslots) return;
        const int2 item = items[blockIdx.x];
        if (item.x < 0) return;
#ifdef MM_LANE_STAMPS
        asm volatile("" ::"s"(item.x));
#endif
        MM_STAMP(0);   // kernel start / previous item -> item descriptor here
        const int col = item.x / nstrips, strip = item.x - col * nstrips;
        const int cx = col / g.ny, cy = col - cx * g.ny;
        const int cz0 = strip * Z, cz1 = min(cz0 + Z, g.nz);
        const int za = max(cz0 - 1, 0), zb = min(cz1, g.nz - 1);
        const int nlayers = zb - za + 1;
        const int ntc = nlayers * 9;                      // <= (kLaneZMax + 2) * 9 = 126 cells: two per lane
        // ---- extents of the tile's cells (cell q = 9 * layer + column), two per lane
        int s0[2] = {0, 0}, cnt[2] = {0, 0};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int q = lane + 64 * b;
            const int layer = q / 9, c = q - layer * 9;
            const int ix = cx + c / 3 - 1, iy = cy + (c - (c / 3) * 3) - 1;
            if (q < ntc && (unsigned)ix < (unsigned)g.nx && (unsigned)iy < (unsigned)g.ny) {
                const int cellid = (ix * g.ny + iy) * g.nz + za + layer;
                s0[b] = cell_start[cellid];
                // (clamped: a cell of a clustered cloud can hold more than the 16 bits the packed prefix sums below
                // give a running total -- one source too many for the tile is all the overflow test needs to see)
                cnt[b] = min(cell_start[cellid + 1] - s0[b], kLaneTileCap + 1);
            }
        }
        int t0 = tstart[col * g.nz + cz0];
        const int t1 = tstart[col * g.nz + cz1];
        t0 += item.y * per_item;
        const int tn = min(per_item, t1 - t0);            // this item's share of the strip's targets
        const double ox = g.lox + (double)cx * g.hx;
        const double oy = g.loy + (double)cy * g.hy;
        const double oz = g.loz + (double)cz0 * g.hz;
        // the first round's targets: in flight while the tile is staged
        double npx, npy, npz, npw;
        {
            const double2 *r2 = reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + (lane < tn ? lane : 0)) * kRec);
            const double2 xy = r2[0], zw = r2[1];
            npx = xy.x;
            npy = xy.y;
            npz = zw.x;
            npw = zw.y;
        }
        // ---- natural tile offsets: prefix sum over the cells in (layer, column) order
        int nat[2], nat_total;
        {
            // (both prefix sums in one word -- a tile holds < 2^16 sources and the counts are clamped --: six dependent
            // shuffles instead of twelve; unsigned, so that the upper sum may use all of its 16 bits)
            const unsigned packed = (unsigned)wave_inclusive_sum((int)((unsigned)cnt[0] | ((unsigned)cnt[1] << 16)));
            const int incl0 = (int)(packed & 0xffffu), incl1 = (int)(packed >> 16);
            const int tot0 = __builtin_amdgcn_readlane(incl0, kWave - 1);
            nat_total = tot0 + __builtin_amdgcn_readlane(incl1, kWave - 1);
            nat[0] = incl0 - cnt[0];
            nat[1] = tot0 + incl1 - cnt[1];
        }
#ifdef MM_LANE_STAMPS
        asm volatile("" ::"v"(nat[0]), "v"(nat[1]));
#endif
        MM_STAMP(1);   // cell extents arrived, offsets computed
        if (nat_total > kLaneTileCap) {
            // too full for the tile (a locally much denser region): the item's targets go to the next density
            // level when there is one (a grid with smaller cells there), else to the generic kernel
            int base = 0;
            if (down_list) {
                if (lane == 0) base = atomicAdd(down_count, tn);
                base = __shfl(base, 0);
                for (int q = lane; q < tn; q += kWave) down_list[base + q] = record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
            } else {
                if (lane == 0) base = atomicAdd(fb_count, tn);
                base = __shfl(base, 0);
                for (int q = lane; q < tn; q += kWave)
                    fb_list[base + q] = sorted_rows ? t0 + q : record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
            }
            return;
        }
        // ---- stage, step 1: every entry's position in the sorted array, in cell order (the cells' owners know them) ...
#pragma unroll
        for (int b = 0; b < 2; ++b)
            for (int q = 0; q < cnt[b]; ++q) reinterpret_cast<int *>(tile + nat[b] + q)[3] = s0[b] + q;
        s_hist[lane] = 0;
        wave_sync();
        MM_STAMP(2);   // positions written
        // ... step 2: entry 64 u + lane is fetched by lane `lane` -- the WHOLE tile sits in registers at once
        // (kLaneTrips records per lane, all in flight together: one global round trip), is converted to fp32
        // relative to the strip corner and BINNED BY THIN LAYER: a cell layer is cut into T slices along z, the
        // tile is kept in (thin layer, arrival) order, and a target's window is the 2 W + 1 thin layers around
        // its own instead of three whole cell layers (7 / 4 of a cell edge instead of 3: 40 % fewer candidates,
        // and candidates are what the scan's vector instructions are spent on).  The rank inside the thin layer
        // comes back from the LDS atomic that counts it.
        const int NL = nlayers * T;                              // <= kLaneThinMax (the launcher checks)
        const float zbase = (float)((double)(za - cz0) * g.hz);  // z of the tile's bottom, relative to the strip corner
        const double th = g.hz / (double)T;                      // thickness of a thin layer
        const float inv_t = (float)((double)T * g.ihz);
        {
            // (everything a lane holds of the tile stays in registers between the two LDS phases: the entries cannot be
            // parked in their cell-order slots, which the final order overwrites)
            int epos[kLaneTrips], ebin[kLaneTrips];   // ebin: thin layer | rank inside it << 8
            float ex[kLaneTrips], ey[kLaneTrips], ez[kLaneTrips];
            {
                double2 xy[kLaneTrips];
                double zc[kLaneTrips];
#pragma unroll
                for (int u = 0; u < kLaneTrips; ++u) {
                    const int e = u * kWave + lane;
                    epos[u] = e < nat_total ? reinterpret_cast<const int *>(tile + e)[3] : -1;
                }
#pragma unroll
                for (int u = 0; u < kLaneTrips; ++u) {
                    const double *rec = sorted_xyz + (i64)max(epos[u], 0) * kRec;
                    xy[u] = *reinterpret_cast<const double2 *>(rec);
                    zc[u] = rec[2];
                }
#pragma unroll
                for (int u = 0; u < kLaneTrips; ++u) {
                    // non-finite or absurdly far sources become far-away entries (never NaN in a key)
                    ex[u] = fminf(fmaxf((float)(xy[u].x - ox), -kLaneFar), kLaneFar);
                    ey[u] = fminf(fmaxf((float)(xy[u].y - oy), -kLaneFar), kLaneFar);
                    ez[u] = fminf(fmaxf((float)(zc[u] - oz), -kLaneFar), kLaneFar);
                    const int tl = min(max((int)((ez[u] - zbase) * inv_t), 0), NL - 1);
                    ebin[u] = tl;
                    if (epos[u] >= 0) ebin[u] = tl | (atomicAdd(&s_hist[tl], 1) << 8);
                }
            }
            wave_sync();   // every entry binned; every position read
            // thin-layer starts: exclusive prefix over the bins, lane = thin layer (bins past NL are empty)
            {
                const int c = s_hist[lane];
                const int incl = wave_inclusive_sum(c);
                s_thin[lane] = incl - c;
                if (lane == kWave - 1) s_thin[kWave] = incl;
            }
            if (lane < kLanePad) tile[nat_total + lane] = far_entry;   // a window read may run past the tile's end
            wave_sync();
#pragma unroll
            for (int u = 0; u < kLaneTrips; ++u)
                if (epos[u] >= 0)
                    tile[s_thin[ebin[u] & 255] + (ebin[u] >> 8)] = make_float4(ex[u], ey[u], ez[u], __int_as_float(epos[u]));
        }
        wave_sync();   // tile and thin-layer table staged
        MM_STAMP(3);   // records gathered, converted, in LDS

        for (int r0 = 0; r0 < tn; r0 += kWave) {
            const bool valid = r0 + lane < tn;
            const double px = npx, py = npy, pz = npz;
            const i64 i = sorted_rows ? (i64)(t0 + r0 + lane) : (i64)record_id(npw);
            if (r0 + kWave < tn) {
                // the next round's targets, in flight during this round
                const int q = r0 + kWave + lane;
                const double2 *r2 = reinterpret_cast<const double2 *>(tsorted + (i64)(t0 + (q < tn ? q : 0)) * kRec);
                const double2 xy = r2[0], zw = r2[1];
                npx = xy.x;
                npy = xy.y;
                npz = zw.x;
                npw = zw.y;
            }
            const bool finite = isfinite(px) && isfinite(py) && isfinite(pz);
            const float tx = finite ? (float)(px - ox) : 0.f, ty = finite ? (float)(py - oy) : 0.f,
                        tz = finite ? (float)(pz - oz) : 0.f;
            // the target's thin layer, by the arithmetic that binned the sources
            const int tlz = min(max((int)((tz - zbase) * inv_t), 0), NL - 1);
            double ed[NE];
            int ei[NE];
            int rank[NE];
            bool hand_over;
            // First the narrow window (W thin layers either way).  When any target of the round cannot be certified
            // in it -- its k-th neighbour is farther than the window's faces, or the window holds too few sources:
            // sparser places than the grid was laid out for -- the scan goes on over what a full cell layer either
            // way adds (the entries above and below what every lane has read already, into the same lists) and the
            // round is certified against that window: the guarantee of a 3x3x3 block.
            int lo = max(tlz - (kRetry ? W : T), 0), hi = min(tlz + (kRetry ? W : T), NL - 1);
            const int we = s_thin[hi + 1];
            int nsteps;
            {
                // the longest window of the round: the trip count of every lane's scan (< 1024: the payload's 10 bits)
                const int wl = wave_max_nonneg(valid ? we - s_thin[lo] : 0);
                nsteps = max((wl + kLaneUnroll - 1) / kLaneUnroll * kLaneUnroll, kLaneUnroll);
            }
            // every lane reads nsteps entries ending at its window's end (or starting at the tile's start)
            const int wbase = max(we - nsteps, 0);
            int ext_hi = 0, ext_lo = 0;   // (wave-uniform) entries read behind / before [wbase, wbase + nsteps) so far
            float d[L];
#pragma unroll
            for (int s = 0; s < L; ++s) d[s] = 3.0e38f;
            const float4 *wp = tile + wbase;
            unsigned key_mask = 0xfffffc00u;
            float neg_inf = -INFINITY;
            asm volatile("" : "+v"(key_mask), "+v"(neg_inf));   // both stay in registers (see lane_list_insert)
            // two half-chunks in flight: the LDS reads of one are issued before the other is consumed
            constexpr int H = kLaneUnroll / 2;
            float4 qa[H], qb[H];
#pragma unroll
            for (int u = 0; u < H; ++u) qa[u] = wp[u];
            for (int j = 0; j < nsteps; j += kLaneUnroll) {
#pragma unroll
                for (int u = 0; u < H; ++u) qb[u] = wp[j + H + u];
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    // the whole entry is asked for: one ds_read_b128 (4 LDS cycles per wave); the 12 bytes alone
                    // come as a ds_read_b96 (8 cycles), split arrays as ds_read2_b64 + ds_read2_b32 (6 per entry)
                    asm volatile("" ::"v"(qa[u].w));
                    const float fx = qa[u].x - tx, fy = qa[u].y - ty, fz = qa[u].z - tz;
                    const float d2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                    // key = (d2 & ~1023) | slot: one v_and_or_b32, the slot (wave-uniform) from a scalar register
                    float key;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(d2), "v"(key_mask), "s"(j + u));
                    lane_list_insert<L>(d, key, neg_inf);
                }
#pragma unroll
                for (int u = 0; u < H; ++u) qa[u] = wp[j + kLaneUnroll + u];   // (past the last chunk: the padding)
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    asm volatile("" ::"v"(qb[u].w));
                    const float fx = qb[u].x - tx, fy = qb[u].y - ty, fz = qb[u].z - tz;
                    const float d2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                    float key;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(d2), "v"(key_mask), "s"(j + H + u));
                    lane_list_insert<L>(d, key, neg_inf);
                }
            }

#ifdef MM_LANE_STAMPS
            asm volatile("" ::"v"(d[0]), "v"(d[L - 1]));
#endif
            MM_STAMP(4);   // scan
            for (int attempt = 0;; ++attempt) {
            // ---- exact fp64 distance (reference arithmetic) and source id of the K + 1 best keys
            {
                int pos[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    // slot -> tile entry: [0, nsteps) the first scan, then the entries behind it, then those before it
                    const int slot = (int)(__float_as_uint(d[e]) & 1023u);
                    const int idx = slot < nsteps + ext_hi ? wbase + slot : wbase - ext_lo + (slot - nsteps - ext_hi);
                    pos[e] = __float_as_int(tile[min((unsigned)idx, (unsigned)nat_total)].w);   // (sentinel keys: any entry)
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const double2 *r2 = reinterpret_cast<const double2 *>(sorted_xyz + (i64)max(pos[e], 0) * kRec);
                    const double2 xy = r2[0], zw = r2[1];
                    const double dx = xy.x - px;
                    const double dy = xy.y - py;
                    const double dz = zw.x - pz;
                    double d2 = dx * dx;
                    d2 = d2 + dy * dy;
                    if (ndim > 2) d2 = d2 + dz * dz;
                    const bool real = d[e] < kLaneFarKey;
                    ed[e] = real ? d2 : INFINITY;
                    ei[e] = real ? record_id(zw.y) : 0x7fffffff - e;   // distinct ids keep sentinels apart
                }
            }
#ifdef MM_LANE_STAMPS
            asm volatile("" ::"v"(ed[0]), "v"(ed[NE - 1]));
#endif
            MM_STAMP(5);   // exact distances here
            double kth = INFINITY;
            if (!kRowsInLds) {
                // Short rows: the candidates come in KEY order, which is the exact order except where two exact
                // distances lie within the keys' resolution (2^-13 relative: a few per cent of the targets have one
                // such pair among their nine).  Adjacent swaps on (d2, id) until every lane's list is in order --
                // usually one pass -- instead of counting 72 ranks and selecting every output slot out of nine.
                for (;;) {
                    bool inorder = true;
#pragma unroll
                    for (int e = 0; e + 1 < NE; ++e) inorder = inorder && !before(ed[e + 1], ei[e + 1], ed[e], ei[e]);
                    if (!__any(valid && !inorder)) break;
#pragma unroll
                    for (int e = 0; e + 1 < NE; ++e) {
                        const bool sw = before(ed[e + 1], ei[e + 1], ed[e], ei[e]);
                        const double da = ed[e], db = ed[e + 1];
                        const int ia = ei[e], ib = ei[e + 1];
                        ed[e] = sw ? db : da;
                        ed[e + 1] = sw ? da : db;
                        ei[e] = sw ? ib : ia;
                        ei[e + 1] = sw ? ia : ib;
                    }
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    rank[e] = e;
                    if (e == kout - 1) kth = ed[e];
                }
            } else {
            // rank by exact d2; bit-equal distances (rare) redo the ranks lexicographically by (d2, id)
            unsigned seen = 0u;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                int rk = 0;
#pragma unroll
                for (int f = 0; f < NE; ++f)
                    if (f != e) rk += ed[f] < ed[e] ? 1 : 0;
                rank[e] = rk;
                seen |= 1u << rk;
            }
            // distinct distances <=> the ranks are a permutation of 0 .. NE-1 (two sentinels tie as well)
            if (__any(valid && seen != (NE >= 32 ? ~0u : ((1u << NE) - 1u)))) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    int rk = 0;
#pragma unroll
                    for (int f = 0; f < NE; ++f)
                        if (f != e) rk += before(ed[f], ei[f], ed[e], ei[e]) ? 1 : 0;
                    rank[e] = rk;
                }
            }
            // the exact k-th distance
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (rank[e] == kout - 1) kth = ed[e];
            }
            hand_over = !finite || !(kth < INFINITY);
            {
                // every source outside the list lies at an exact distance >= lb (header comment)
                const double E = 3.0 * kU * (fabs(px - ox) + fabs(py - oy) + fabs(pz - oz) + 2.0 * (g.hx + g.hy) +
                                             (double)(Z + 1) * g.hz);
                const float B = d[L - 1];
                if (B < kLaneFarKey) {
                    // (v_sqrt_f32 is within 1 ulp: 2^-21 more off the factor covers it)
                    const double lb = ((double)__builtin_amdgcn_sqrtf(B) * (1.0 - 0x1p-12 - 0x1p-21) - E) * (1.0 - 4.0 * kU);
                    if (!(lb > 0.0 && kth < lb * lb * (1.0 - 0x1p-40))) hand_over = true;
                }
                // could a nearer source sit outside what was scanned?  Beyond the x / y faces of the 3 x 3 columns
                // (as block_bound: faces that still have cells behind them) ...
                double bound = INFINITY;
                const double slack_x = 1e-9 * g.hx, slack_y = 1e-9 * g.hy;
                if (cx - 1 > 0) bound = fmin(bound, (px - (g.lox + (double)(cx - 1) * g.hx)) - slack_x);
                if (cx + 1 < g.nx - 1) bound = fmin(bound, ((g.lox + (double)(cx + 2) * g.hx) - px) - slack_x);
                if (cy - 1 > 0) bound = fmin(bound, (py - (g.loy + (double)(cy - 1) * g.hy)) - slack_y);
                if (cy + 1 < g.ny - 1) bound = fmin(bound, ((g.loy + (double)(cy + 2) * g.hy) - py) - slack_y);
                // ... or in a thin layer below / above the window.  The planes between thin layers are taken in the
                // tile's fp32 frame, where the sources were binned: a source outside the window has an fp32 z beyond
                // the plane (the bin arithmetic is off by < 1e-5 of a thin layer), its coordinate and the target's
                // are within E of the exact ones.  A clipped window ends at the tile's own face, which has sources
                // behind it unless it is the grid's.
                const double zs = E + 1e-4 * th;
                if (lo > 0 || za > 0) bound = fmin(bound, ((double)tz - ((double)zbase + (double)lo * th)) - zs);
                if (hi < NL - 1 || zb < g.nz - 1) bound = fmin(bound, (((double)zbase + (double)(hi + 1) * th) - (double)tz) - zs);
                if (bound < INFINITY && !(bound > 0.0 && kth < bound * bound)) hand_over = true;
            }
            if (!kRetry || attempt > 0 || W >= T || !__any(valid && hand_over)) break;
            // ---- widen to a full cell layer either way: scan what that adds to each side of the entries already read
            lo = max(tlz - T, 0);
            hi = min(tlz + T, NL - 1);
            {
                const int need_hi = wave_max_nonneg(valid ? max(s_thin[hi + 1] - (wbase + nsteps), 0) : 0);
                const int need_lo = wave_max_nonneg(valid ? max(wbase - s_thin[lo], 0) : 0);
                ext_hi = (need_hi + kLaneUnroll - 1) / kLaneUnroll * kLaneUnroll;
                ext_lo = (need_lo + kLaneUnroll - 1) / kLaneUnroll * kLaneUnroll;
            }
            if (nsteps + ext_hi + ext_lo > 1023) {   // (the payload has 10 bits: such a tile's targets are handed over)
                ext_hi = ext_lo = 0;
                break;
            }
            // (entries beyond the tile's ends read as its far-away padding entry)
            for (int j = 0; j < ext_hi + ext_lo; j += kLaneUnroll) {
                const int first = j < ext_hi ? wbase + nsteps + j : wbase - ext_lo + (j - ext_hi);
                float4 q[kLaneUnroll];
#pragma unroll
                for (int u = 0; u < kLaneUnroll; ++u) q[u] = tile[min((unsigned)(first + u), (unsigned)nat_total)];
#pragma unroll
                for (int u = 0; u < kLaneUnroll; ++u) {
                    const float fx = q[u].x - tx, fy = q[u].y - ty, fz = q[u].z - tz;
                    const float d2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                    float key;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(d2), "v"(key_mask), "s"(nsteps + j + u));
                    lane_list_insert<L>(d, key, neg_inf);
                }
            }
            }   // (attempts)
            if (dist_out && valid && !hand_over) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (rank[e] < kout) dist_out[i * kout + rank[e]] = sqrt(ed[e]);
            }
            if (kRowsInLds) {
                // rows in rank order through LDS, then wide stores
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if (rank[e] < kout) s_row[lane][rank[e]] = ei[e];
                wave_sync();
                if (valid && !hand_over) {
                    IDX *row = idx_out + i * kout;
                    if (sizeof(IDX) == 4 && (kout & 3) == 0) {
                        for (int e = 0; e < kout; e += 4)
                            *reinterpret_cast<int4 *>(row + e) =
                                make_int4(s_row[lane][e], s_row[lane][e + 1], s_row[lane][e + 2], s_row[lane][e + 3]);
                    } else if (sizeof(IDX) == 8 && (kout & 1) == 0) {
                        for (int e = 0; e < kout; e += 2)
                            *reinterpret_cast<longlong2 *>(row + e) = make_longlong2((i64)s_row[lane][e], (i64)s_row[lane][e + 1]);
                    } else {
                        for (int e = 0; e < kout; ++e) row[e] = (IDX)s_row[lane][e];
                    }
                }
            } else {
                // short rows: entry r IS rank r (put in order above)
                int out[K];
#pragma unroll
                for (int r = 0; r < K; ++r) out[r] = ei[r];
                if (valid && !hand_over) {
                    IDX *row = idx_out + i * kout;
                    if (sizeof(IDX) == 4 && K % 4 == 0 && kout == K) {
#pragma unroll
                        for (int e = 0; e < K; e += 4)
                            *reinterpret_cast<int4 *>(row + e) = make_int4(out[e], out[e + 1], out[e + 2], out[e + 3]);
                    } else if (sizeof(IDX) == 8 && K % 2 == 0 && kout == K) {
#pragma unroll
                        for (int e = 0; e < K; e += 2)
                            *reinterpret_cast<longlong2 *>(row + e) = make_longlong2((i64)out[e], (i64)out[e + 1]);
                    } else {
#pragma unroll
                        for (int e = 0; e < K; ++e)
                            if (e < kout) row[e] = (IDX)out[e];
                    }
                }
            }
            // hand-overs of this round: one atomic per wave
            const unsigned long long mf = __ballot(valid && hand_over);
            if (mf) {
                const int firstl = __ffsll((long long)mf) - 1;
                int base = 0;
                if (lane == firstl) base = atomicAdd(fb_count, __popcll(mf));
                base = __shfl(base, firstl);
                if (valid && hand_over) fb_list[base + __popcll(mf & ((1ull << lane) - 1ull))] = (int)i;
            }
            if (kRowsInLds) wave_sync();   // rows are rewritten by the next round
            MM_STAMP(6);   // ranks, certification, output
        }
    }
#ifdef MM_LANE_STAMPS
    if (lane == 0) {
        unsigned long long *slot = g_lane_stamps + (size_t)(blockIdx.x & (kStampSlots - 1)) * 8;
        for (int q = 0; q < 7; ++q) slot[q] += stamp_sum[q];   // (grids beyond the slot count alias: sums only)
        slot[7] += 1ull;
    }
#endif
}

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

void free_index(mm_knn_index *ix)
{
    if (!ix) return;
    free_index(ix->fine);
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

// Build without touching the stage timers (used by the fused pipeline too).
int mm_knn_build_impl(mm_context *ctx, const double *src_d, i64 nsrc, i64 ndim, mm_knn_index **out,
                      bool use_context_buffers, const double *box_partial_d, int box_nblocks)
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

// tsorted_out (nullable): the caller can take the rows in the cell-sorted order of the targets; on return
// *tsorted_out = the sorted target records {x, y, z, index} (context buffer, valid until the next query) when the
// rows were written in that order, null when they are in the targets' own order (paths without the lane kernel).
template <typename IDX>
static int knn_query_typed(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, IDX *idx_d,
                           double *dist_d, const double **tsorted_out, const int *list0 = nullptr,
                           const int *list0_count = nullptr)
{
    // list0 (device): only the targets list0[0 .. *list0_count) are served (rows by the targets' own indices as ever);
    // the tiled kernels then walk just the strips that hold any of them (k <= 32)
    if (tsorted_out) *tsorted_out = nullptr;
    if (npts == 0 || k == 0) return MM_OK;
    MM_REQUIRE(npts < (i64)0x7fffffff, "too many targets for one query");
    const GridParams g = params_of(ix);
    const int kout = (int)k;
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
        if (!(level == 0 && one_fill))
            if (mm_zero_async(ctx, counts, mm_fill_span((size_t)(ncells + 1) * sizeof(int))) != MM_OK) return MM_ERR_HIP;
        if (!list && ncells <= kHistCells && npts >= 64 * ncells)   // many targets over few cells (see the kernel)
            hipLaunchKernelGGL(cell_count_hist_kernel, dim3(256), dim3(kHistBlock), 0, ctx->stream, pts_d, npts, l->ndim, gl,
                               (int)ncells, cell_of, counts);
        else
            hipLaunchKernelGGL(cell_count_kernel, dim3(gpts), dim3(kBlock), 0, ctx->stream, pts_d, npts, l->ndim, gl, cell_of,
                               counts, list, list_count);
        hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(kBlock), 0, ctx->stream, counts, ncells, tile_sums,
                           (unsigned long long *)nullptr, 0);
        launch_scan_tail(ctx, counts, ncells, tile_sums, ntiles, start);
        hipLaunchKernelGGL(target_scatter_kernel, dim3(gpts), dim3(kBlock), 0, ctx->stream, cell_of, npts, pts_d, l->ndim, gl,
                           start, tsorted, list, list_count);
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
    if (list_len_hint >= kLongListMin && k <= 32) return knn_query_typed<int>(ctx, ix, pts_d, npts, k, idx_d, nullptr, nullptr, list, list_count);
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
    int rc = mm_knn_build_impl(ctx, src_d, nsrc, ndim, out, false, nullptr, 0);
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
