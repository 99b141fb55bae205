// A2, the lane-per-target kernel over an LDS tile ordered by thin layers (the metric's kNN kernel) and its work-item prepass.
// A part of mm_knn.hip -- ONE translation unit: the kernels of all parts are instantiated from its launchers --, included
// inside that file's anonymous namespace in the order grid, rings, tiles, lane.  Not a header to include elsewhere.

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

// TREE: the work items, their windows and the faces of the certificate come from the density-adaptive index
// (mm_knn_tree.inc.h) instead of the uniform grid g; everything from the staged tile on is the same code.
template <int K, typename IDX, bool TREE = false>
__global__ __launch_bounds__(kWave, 3) void knn_lane_kernel(GridParams g, i64 nsrc, const int *__restrict__ cell_start,
                                                            const double *__restrict__ sorted_xyz, int ndim, int kout,
                                                            const int *__restrict__ tstart,
                                                            const double *__restrict__ tsorted, IDX *__restrict__ idx_out,
                                                            double *__restrict__ dist_out, int *__restrict__ fb_list,
                                                            int *__restrict__ fb_count, const int2 *__restrict__ items,
                                                            int nslots, int Z, int per_item, int sorted_rows,
                                                            int *__restrict__ down_list, int *__restrict__ down_count,
                                                            int T, int W, TreeArgs ta = TreeArgs())
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
        // (tree_* : the window of a tree item -- cells of level tree_lc and edge tree_c, tree_nax/y/z per axis, the first one at
        // (tree_bx, tree_by, tree_bz); unused otherwise)
        int tree_lc = 0, tree_nax = 0, tree_nay = 0, tree_naz = 0, tree_bx = 0, tree_by = 0, tree_bz = 0;
        double tree_c = 0.0;
        int cx = 0, cy = 0, cz0 = 0, za = 0, zb = 0, nlayers, t0, tn;
        constexpr int NB = TREE ? 4 : 2;   // cells per lane (6^3 = 216 cells in the largest tree window)
        int s0[NB], cnt[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) s0[b] = cnt[b] = 0;
        double ox, oy, oz;
        if (TREE) {
            // item q of the list (Morton order) is taken by workgroup 8 m + x, x = the eighth of the list it lies in: the
            // slot rule of the uniform kernel, worked out here from the item count on the device
            const int total = *ta.nitems;
            const int x8 = (int)blockIdx.x & 7, m8 = (int)blockIdx.x >> 3;
            const i64 q0 = ((i64)total * x8) >> 3, q1 = ((i64)total * (x8 + 1)) >> 3;
            if (q0 + m8 >= q1) return;
            const TreeItem it = ta.items[q0 + m8];
            t0 = it.t0;
            tn = ta.items[q0 + m8 + 1].t0 - t0;
            // the node: P leading key bits -- the octree cell of level P / 3, halved along x (P % 3 >= 1) and y (P % 3 == 2);
            // its window: one cell of level P / 3 + d around it (mm_knn_tree.inc.h)
            const int P = (int)((it.node >> kTreeLevelShift) & 0x3f), deeper = (int)(it.node >> 62);
            const u64 k0n = it.node & ((1ull << kTreeLevelShift) - 1ull);
            tree_lc = P / 3 + deeper;
            const int shq = kTreeQ - tree_lc;
            // (the node's first key is its corner: the low bits are zero)
            tree_bx = (int)(tree_key_x(k0n) >> shq) - 1;
            tree_by = (int)(tree_key_y(k0n) >> shq) - 1;
            tree_bz = (int)(tree_key_z(k0n) >> shq) - 1;
            const int jj = P - 3 * (P / 3), aa = 1 << deeper;
            tree_nax = (jj >= 1 ? aa >> 1 : aa) + 2;
            tree_nay = (jj >= 2 ? aa >> 1 : aa) + 2;
            tree_naz = aa + 2;
            tree_c = ta.tp.size / (double)(1 << tree_lc);
            nlayers = tree_naz;
            // ---- extents of the window's cells, up to four per lane: two searches on the sorted keys each
            const int ncl = 1 << tree_lc;
            const int sh = 3 * shq;
            const int wcells = tree_nax * tree_nay * tree_naz;
            // (all of a lane's searches advance together, one step per trip: their loads are in flight at once -- done one
            // after the other, eight searches of ~6 dependent loads each were half of an item's time)
            u64 kq[2 * NB];
            int blo[2 * NB], bhi[2 * NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int q = lane + 64 * b;
                const int iz = q / (tree_nax * tree_nay), r2 = q - iz * tree_nax * tree_nay, iy = r2 / tree_nax, ix = r2 - iy * tree_nax;
                const int gx = tree_bx + ix, gy = tree_by + iy, gz = tree_bz + iz;
                const bool in = q < wcells && (unsigned)gx < (unsigned)ncl && (unsigned)gy < (unsigned)ncl && (unsigned)gz < (unsigned)ncl;
                const u64 k0 = tree_morton((unsigned)(in ? gx : 0), (unsigned)(in ? gy : 0), (unsigned)(in ? gz : 0)) << sh;
                kq[2 * b] = k0;
                kq[2 * b + 1] = k0 + (1ull << sh);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u64 key = kq[2 * b + e];
                    const bool past = (key >> kTreeBits) != 0;
                    const int c = past ? 0 : (int)(key >> kTreeShift0);
                    const int a0 = ta.coarse[c], a1 = ta.coarse[c + 1];
                    blo[2 * b + e] = !in ? 0 : (past ? (int)nsrc : a0);
                    bhi[2 * b + e] = !in ? 0 : (past ? (int)nsrc : a1);
                }
            }
            for (;;) {
                u64 kv[2 * NB];
                int mid[2 * NB];
#pragma unroll
                for (int j = 0; j < 2 * NB; ++j) {
                    mid[j] = (blo[j] + bhi[j]) >> 1;
                    kv[j] = ta.keys[min(mid[j], (int)nsrc - 1)];
                }
                bool more = false;
#pragma unroll
                for (int j = 0; j < 2 * NB; ++j) {
                    if (blo[j] < bhi[j]) {
                        if (kv[j] < kq[j]) blo[j] = mid[j] + 1;
                        else bhi[j] = mid[j];
                    }
                    more = more || blo[j] < bhi[j];
                }
                if (!__any(more)) break;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                s0[b] = blo[2 * b];
                cnt[b] = min(blo[2 * b + 1] - blo[2 * b], kLaneTileCap + 1);
            }
            // (the tile's frame: the corner of the cell behind the window's first one, i.e. the node's own corner)
            ox = ta.tp.lox + (double)(tree_bx + 1) * tree_c;
            oy = ta.tp.loy + (double)(tree_by + 1) * tree_c;
            oz = ta.tp.loz + (double)(tree_bz + 1) * tree_c;
        } else {
        if ((int)blockIdx.x >= nslots) return;
        const int2 item = items[blockIdx.x];
        if (item.x < 0) return;
#ifdef MM_LANE_STAMPS
        asm volatile("" ::"s"(item.x));
#endif
        MM_STAMP(0);   // kernel start / previous item -> item descriptor here
        const int col = item.x / nstrips, strip = item.x - col * nstrips;
        cx = col / g.ny;
        cy = col - cx * g.ny;
        cz0 = strip * Z;
        const int cz1 = min(cz0 + Z, g.nz);
        za = max(cz0 - 1, 0);
        zb = min(cz1, g.nz - 1);
        nlayers = zb - za + 1;
        const int ntc = nlayers * 9;                      // <= (kLaneZMax + 2) * 9 = 126 cells: two per lane
        // ---- extents of the tile's cells (cell q = 9 * layer + column), two per lane
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
        t0 = tstart[col * g.nz + cz0];
        const int t1 = tstart[col * g.nz + cz1];
        t0 += item.y * per_item;
        tn = min(per_item, t1 - t0);            // this item's share of the strip's targets
        ox = g.lox + (double)cx * g.hx;
        oy = g.loy + (double)cy * g.hy;
        oz = g.loz + (double)cz0 * g.hz;
        }
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
        int nat[NB], nat_total;
        if (TREE) {
            int lane_total = 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) lane_total += cnt[b];
            const int incl = wave_inclusive_sum(lane_total);
            nat_total = __builtin_amdgcn_readlane(incl, kWave - 1);
            nat[0] = incl - lane_total;
#pragma unroll
            for (int b = 1; b < NB; ++b) nat[b] = nat[b - 1] + cnt[b - 1];
        } else {
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
        if (TREE) nat_total = min(nat_total, kLaneTileCap + 1);   // (the sum of clamped counts: any overflow reads as one)
        MM_STAMP(1);   // cell extents arrived, offsets computed
        if (nat_total > kLaneTileCap) {
            // too full for the tile (a locally much denser region): the item's targets go to the next density
            // level when there is one (a grid with smaller cells there), else to the generic kernel
            int base = 0;
            if (down_list) {
                if (lane == 0) base = atomicAdd(down_count, tn);
                base = __shfl(base, 0);
                for (int q = lane; q < tn; q += kWave)
                    down_list[base + q] = TREE && sorted_rows ? t0 + q : record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
            } else {
                if (lane == 0) base = atomicAdd(fb_count, tn);
                if (TREE && lane == 0) atomicAdd(fb_count + 2, tn);   // (diagnostic: hand-overs by overflow)
                base = __shfl(base, 0);
                for (int q = lane; q < tn; q += kWave)
                    fb_list[base + q] = sorted_rows ? t0 + q : record_id(tsorted[(i64)(t0 + q) * kRec + 3]);
            }
            return;
        }
        // ---- stage, step 1: every entry's position in the sorted array, in cell order (the cells' owners know them) ...
#pragma unroll
        for (int b = 0; b < NB; ++b)
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
        const float zbase = TREE ? (float)(-tree_c) : (float)((double)(za - cz0) * g.hz);  // z of the tile's bottom, relative to the strip corner
        const double th = (TREE ? tree_c : g.hz) / (double)T;    // thickness of a thin layer
        const float inv_t = TREE ? (float)((double)T / tree_c) : (float)((double)T * g.ihz);
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
                const double E = 3.0 * kU * (fabs(px - ox) + fabs(py - oy) + fabs(pz - oz) +
                                             (TREE ? (double)(tree_nax + tree_nay + tree_naz) * tree_c
                                                   : 2.0 * (g.hx + g.hy) + (double)(Z + 1) * g.hz));
                const float B = d[L - 1];
                if (B < kLaneFarKey) {
                    // (v_sqrt_f32 is within 1 ulp: 2^-21 more off the factor covers it)
                    const double lb = ((double)__builtin_amdgcn_sqrtf(B) * (1.0 - 0x1p-12 - 0x1p-21) - E) * (1.0 - 4.0 * kU);
                    if (!(lb > 0.0 && kth < lb * lb * (1.0 - 0x1p-40))) hand_over = true;
                }
                // could a nearer source sit outside what was scanned?  Beyond the x / y faces of the 3 x 3 columns
                // (as block_bound: faces that still have cells behind them) ...
                double bound = INFINITY;
                bool below, above;   // cells under / over the tile
                if (TREE) {
                    // the window's faces in x and y (those with cells of the cube behind them); a source of a cell beyond a
                    // face has a coordinate beyond it up to the rounding of tree_quant's product (< 1e-10 of a cell edge)
                    const double slack = 1e-9 * tree_c;
                    const int ncl = 1 << tree_lc;
                    if (tree_bx > 0) bound = fmin(bound, (px - (ta.tp.lox + (double)tree_bx * tree_c)) - slack);
                    if (tree_bx + tree_nax < ncl) bound = fmin(bound, ((ta.tp.lox + (double)(tree_bx + tree_nax) * tree_c) - px) - slack);
                    if (tree_by > 0) bound = fmin(bound, (py - (ta.tp.loy + (double)tree_by * tree_c)) - slack);
                    if (tree_by + tree_nay < ncl) bound = fmin(bound, ((ta.tp.loy + (double)(tree_by + tree_nay) * tree_c) - py) - slack);
                    below = tree_bz > 0;
                    above = tree_bz + tree_naz < ncl;
                } else {
                const double slack_x = 1e-9 * g.hx, slack_y = 1e-9 * g.hy;
                if (cx - 1 > 0) bound = fmin(bound, (px - (g.lox + (double)(cx - 1) * g.hx)) - slack_x);
                if (cx + 1 < g.nx - 1) bound = fmin(bound, ((g.lox + (double)(cx + 2) * g.hx) - px) - slack_x);
                if (cy - 1 > 0) bound = fmin(bound, (py - (g.loy + (double)(cy - 1) * g.hy)) - slack_y);
                if (cy + 1 < g.ny - 1) bound = fmin(bound, ((g.loy + (double)(cy + 2) * g.hy) - py) - slack_y);
                below = za > 0;
                above = zb < g.nz - 1;
                }
                // ... or in a thin layer below / above the window.  The planes between thin layers are taken in the
                // tile's fp32 frame, where the sources were binned: a source outside the window has an fp32 z beyond
                // the plane (the bin arithmetic is off by < 1e-5 of a thin layer), its coordinate and the target's
                // are within E of the exact ones.  A clipped window ends at the tile's own face, which has sources
                // behind it unless it is the grid's.
                const double zs = E + 1e-4 * th;
                if (lo > 0 || below) bound = fmin(bound, ((double)tz - ((double)zbase + (double)lo * th)) - zs);
                if (hi < NL - 1 || above) bound = fmin(bound, (((double)zbase + (double)(hi + 1) * th) - (double)tz) - zs);
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
                // (a tree query's first pass: to the second pass, whose windows are laid out with wider margins)
                int *const out_list = TREE && down_list ? down_list : fb_list;
                int *const out_count = TREE && down_list ? down_count : fb_count;
                if (lane == firstl) base = atomicAdd(out_count, __popcll(mf));
                base = __shfl(base, firstl);
                if (valid && hand_over) out_list[base + __popcll(mf & ((1ull << lane) - 1ull))] = (int)i;
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
