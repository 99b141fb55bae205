// A2, search grid: constants, bounding box, cell assignment, the counting sorts' count / scan / scatter kernels, records.
// A part of mm_knn.hip -- ONE translation unit: the kernels of all parts are instantiated from its launchers --, included
// inside that file's anonymous namespace in the order grid, rings, tiles, lane.  Not a header to include elsewhere.


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
        // (the bins of a workgroup are added up in LDS and leave it as ONE atomic per bin: on a graded cloud nearly every wave
        // has cells above several thresholds, and 40 k atomics on eight addresses made this kernel 0.17 ms instead of 0.01)
        __shared__ unsigned s_bin[kMaxLevels];
        if (threadIdx.x < kMaxLevels) s_bin[threadIdx.x] = 0u;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < kMaxLevels - 1; ++b) {
            int w = 0;
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) w += v[i] > kLevelCount[b] ? v[i] : 0;
            if (!__any(w > 0)) break;   // thresholds ascend
            for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
            if ((threadIdx.x & 63) == 0 && w > 0) atomicAdd(&s_bin[b], (unsigned)w);
        }
        __syncthreads();
        if (threadIdx.x < kMaxLevels - 1 && s_bin[threadIdx.x] > 0u)
            atomicAdd(level_total + threadIdx.x, (unsigned long long)s_bin[threadIdx.x]);
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
