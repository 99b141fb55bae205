// A2, ring searches: the generic lane-per-target kernel, its density-level form and the wave-per-target list kernel (stragglers, long lists, k > 32).
// A part of mm_knn.hip -- ONE translation unit: the kernels of all parts are instantiated from its launchers --, included
// inside that file's anonymous namespace in the order grid, rings, tiles, lane.  Not a header to include elsewhere.

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
