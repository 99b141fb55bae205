// A2, round-1 tiled kernels: one wave per cell (flat / 2-D grids) and per strip of two cells (density levels, sparse targets).
// A part of mm_knn.hip -- ONE translation unit: the kernels of all parts are instantiated from its launchers --, included
// inside that file's anonymous namespace in the order grid, rings, tiles, lane.  Not a header to include elsewhere.

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
