// A11 -- unique target points with the index array that rebuilds the input.
// Replaces np.unique(points.reshape(-1, dim), axis=0, return_inverse=True) at reference
// multi_mesh/utils.py:484-488 (the pre-step of every GLL-target flow: element-nodal points repeat on
// shared faces, edges and corners, ~1/3 to 1/2 of them are duplicates; the scatter-back is
// interpolator.py:823).
//
// Rows are ordered lexicographically by (x, y, z) in fp64 comparison order, equal rows collapse to
// one, inverse[i] = position of row i in the unique list.  -0.0 compares equal to +0.0 as in NumPy;
// the row kept for a group of equal rows is the one with the smallest original index (NumPy's
// unstable sort leaves that choice open when rows differ only in the sign of a zero).  NaN
// coordinates are not supported (they are ordered by bit pattern and never merge).
//
// How.  Mesh points are in general position along x except for the copies of shared nodes, so ONE
// stable sort of (order-preserving 64-bit image of x, row index) almost finishes the job: what is left
// are short runs of rows with bit-equal x (the 2 / 4 / 8 copies of a face / edge / corner node), put in
// (y, z, index) order by the thread that finds the run's head.  A cloud with long runs (an unjittered
// structured mesh: x takes n values, runs of n^2 rows) is detected by that same kernel and redone as dim
// stable sorts, least significant coordinate first.  The sort is a hand-written LSD radix sort, 8 bits a
// pass (radix_hist_kernel -> exclusive scan of the [digit][tile] counts -> radix_scatter_kernel); a
// workgroup ranks its tile of 4096 keys in order -- per wave and batch of 64 keys: the lanes with the same
// digit by eight ballots, their rank by a popcount, the wave's running offsets of the 256 digits in LDS --
// so the scatter is stable.  Then head flags / prefix sum / scatter of the unique rows and the inverse.
#include <algorithm>
#include <cstring>
#include <stdlib.h>
#include <vector>

#include "mm_common.h"

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums);

namespace {

constexpr int kBlock = 256;
constexpr int kScanTileItems = 1024;  // items per block of mm_exclusive_scan_int
constexpr int kWave = 64;
constexpr int kBins = 256;            // 8 bits a pass
// The main sort of mm_unique_points orders the rows by the top 64 - kKeyLowBits bits of x's sortable image (sign, exponent,
// 36 bits of mantissa: six passes instead of eight); rows that agree in those bits form a "run" whatever the rest of x says,
// and the runs are put in full (x, y, z, index) order afterwards -- as the runs of bit-equal x (shared nodes) always were.
// Two different x within 1.5e-11 of each other (relative) merely share a run.
constexpr int kKeyLowBits = 16;
constexpr int kSortWaves = 4;
constexpr int kSortBlock = kSortWaves * kWave;
constexpr int kItems = 16;            // keys per thread: batches of 64 consecutive keys per wave
constexpr int kTile = kSortBlock * kItems;
constexpr int kMaxRun = 32;           // longest run of equal x the fix-up kernel sorts in place
constexpr int kMaxLongRuns = 4096;    // longer runs handled as a sub-sort; more of them: the general path

typedef unsigned long long u64;

// order-preserving map double -> uint64 (-0.0 first folded into +0.0)
__device__ __forceinline__ u64 sortable(double v)
{
    if (v == 0.0) v = 0.0;
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(kBlock) void iota_kernel(unsigned *__restrict__ idx, i64 n)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (unsigned)i;
}

// key of coordinate `comp` of the rows in their current order
__global__ __launch_bounds__(kBlock) void key_kernel(const double *__restrict__ pts, i64 n, int dim, int comp,
                                                     const unsigned *__restrict__ order, u64 *__restrict__ key)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) key[j] = sortable(pts[(i64)order[j] * dim + comp]);
}

// ---- LSD radix sort, one pass = hist -> scan -> scatter ------------------------------------------
// counts[digit * ntiles + tile]: in that order the exclusive scan is the output offset of the tile's first
// key with that digit
__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(const u64 *__restrict__ key, i64 n, int shift, int ntiles,
                                                                int *__restrict__ counts)
{
    __shared__ int s_hist[kBins];
    for (int t = threadIdx.x; t < kBins; t += kSortBlock) s_hist[t] = 0;
    __syncthreads();
    const i64 base = (i64)blockIdx.x * kTile;
#pragma unroll 4
    for (int b = 0; b < kItems; ++b) {
        const i64 j = base + (i64)b * kSortBlock + threadIdx.x;   // (any order: this is only a count)
        if (j < n) atomicAdd(&s_hist[(int)((key[j] >> shift) & (kBins - 1))], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kBins; t += kSortBlock) counts[(i64)t * ntiles + blockIdx.x] = s_hist[t];
}

__global__ __launch_bounds__(kSortBlock) void radix_scatter_kernel(const u64 *__restrict__ key_in,
                                                                   const unsigned *__restrict__ val_in, i64 n, int shift,
                                                                   int ntiles, const int *__restrict__ offsets,
                                                                   u64 *__restrict__ key_out,
                                                                   unsigned *__restrict__ val_out)
{
    // The tile is first put in digit order in LDS (stable), then written out: consecutive threads write
    // consecutive addresses inside a digit's run (~16 keys = 128 bytes at 256 digits per 4096 keys) -- scattering
    // straight from registers costs a cache line per 8-byte key in the passes over the mantissa's random bits.
    __shared__ u64 s_key[kTile];
    __shared__ unsigned s_val[kTile];
    __shared__ int s_off[kSortWaves][kBins];   // first the waves' digit counts, then their running positions in the tile
    __shared__ int s_start[kBins];             // where a digit's run starts in the sorted tile
    __shared__ int s_gbase[kBins];             // ... and in the output
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < kSortWaves * kBins; t += kSortBlock) (&s_off[0][0])[t] = 0;
    __syncthreads();
    // wave w owns keys [w * 1024, (w + 1) * 1024) of the tile, batch b = 64 consecutive keys: batch order, then
    // lane order, is the input order
    const i64 tbase = (i64)blockIdx.x * kTile;
    const i64 wbase = tbase + (i64)wave * (kItems * kWave);
    u64 k[kItems];
    unsigned v[kItems];
#pragma unroll
    for (int b = 0; b < kItems; ++b) {
        const i64 j = wbase + b * kWave + lane;
        k[b] = j < n ? key_in[j] : 0;
        v[b] = j < n ? val_in[j] : 0;
        if (j < n) atomicAdd(&s_off[wave][(int)((k[b] >> shift) & (kBins - 1))], 1);
    }
    __syncthreads();
    // digit t (one thread each): its count in the tile; exclusive prefix over the digits by the first wave
    int total = 0;
    if (threadIdx.x < kBins) {
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) total += s_off[w][threadIdx.x];
        s_start[threadIdx.x] = total;
        s_gbase[threadIdx.x] = offsets[(i64)threadIdx.x * ntiles + blockIdx.x];
    }
    __syncthreads();
    if (wave == 0) {
        // 256 counts, four per lane
        int c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[q] = s_start[4 * lane + q];
            sum += c[q];
        }
        int incl = sum;
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            const int t = __shfl_up(incl, dlt);
            if (lane >= dlt) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s_start[4 * lane + q] = run;
            run += c[q];
        }
    }
    __syncthreads();
    if (threadIdx.x < kBins) {
        int run = s_start[threadIdx.x];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const int c = s_off[w][threadIdx.x];
            s_off[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int b = 0; b < kItems; ++b) {
        const i64 j = wbase + b * kWave + lane;
        const bool active = j < n;
        const int d = (int)((k[b] >> shift) & (kBins - 1));
        // the lanes of this batch that hold the same digit
        u64 same = __ballot(active);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const u64 vote = __ballot((d >> bit) & 1);
            same &= ((d >> bit) & 1) ? vote : ~vote;
        }
        const int rank = __popcll(same & lt);
        const int off = active ? s_off[wave][d] : 0;   // (read by every lane of the group before its head moves it on)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active && rank == 0) s_off[wave][d] = off + __popcll(same);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) {
            s_key[off + rank] = k[b];
            s_val[off + rank] = v[b];
        }
    }
    __syncthreads();
    const int valid = (int)(n - tbase < kTile ? n - tbase : kTile);
#pragma unroll 4
    for (int t = threadIdx.x; t < valid; t += kSortBlock) {
        const u64 key = s_key[t];
        const int d = (int)((key >> shift) & (kBins - 1));
        const i64 pos = (i64)s_gbase[d] + (t - s_start[d]);
        key_out[pos] = key;
        val_out[pos] = s_val[t];
    }
}

// Runs of bit-equal keys in the sorted order: the thread at a run's head puts its rows in (y, z, index)
// order (insertion sort on the row indices; runs of shared-node copies are 2 to 8 long); a run longer than
// kMaxRun is only reported.
// (from the x coordinate on: the main sort orders by the TOP bits of x only -- kKeyLowBits below -- so two rows of a run
// may differ in the rest of x)
__device__ __forceinline__ bool row_before(const double *__restrict__ pts, int dim, unsigned a, unsigned b)
{
    for (int c = 0; c < dim; ++c) {
        const u64 ka = sortable(pts[(i64)a * dim + c]), kb = sortable(pts[(i64)b * dim + c]);
        if (ka != kb) return ka < kb;
    }
    return a < b;
}

__global__ __launch_bounds__(kBlock) void run_fixup_kernel(const u64 *__restrict__ key, i64 n, const double *__restrict__ pts,
                                                           int dim, unsigned *__restrict__ order, int *__restrict__ long_runs)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || j + 1 >= n) return;
    const u64 kj = key[j] >> kKeyLowBits;
    if ((j > 0 && (key[j - 1] >> kKeyLowBits) == kj) || (key[j + 1] >> kKeyLowBits) != kj) return;   // not the head of a run of two or more
    int len = 2;
    while (j + len < n && len <= kMaxRun && (key[j + len] >> kKeyLowBits) == kj) ++len;
    if (len > kMaxRun) {
        // (a face of an axis-aligned box mesh: thousands of rows share x) its start goes on a list; the run is
        // sorted with the other long ones afterwards
        const int r = atomicAdd(long_runs, 1);
        if (r < kMaxLongRuns) long_runs[1 + r] = (int)j;
        return;
    }
    unsigned r[kMaxRun];
    for (int t = 0; t < len; ++t) r[t] = order[j + t];
    for (int t = 1; t < len; ++t) {
        const unsigned x = r[t];
        int q = t - 1;
        while (q >= 0 && row_before(pts, dim, x, r[q])) {
            r[q + 1] = r[q];
            --q;
        }
        r[q + 1] = x;
    }
    for (int t = 0; t < len; ++t) order[j + t] = r[t];
}

// length of the run of equal keys that starts at start[r] (the keys are sorted: an upper bound by bisection)
__global__ __launch_bounds__(kBlock) void run_length_kernel(const u64 *__restrict__ key, i64 n, const int *__restrict__ start,
                                                            int nruns, int *__restrict__ len)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const i64 s0 = start[r];
    const u64 k = key[s0] >> kKeyLowBits;
    i64 lo = s0, hi = n;   // key[lo] == k, key[hi] > k (or hi == n), by the bits the main sort ordered
    while (hi - lo > 1) {
        const i64 mid = (lo + hi) >> 1;
        if ((key[mid] >> kKeyLowBits) == k) lo = mid; else hi = mid;
    }
    len[r] = (int)(hi - s0);
}

// table[r] = {start in the sorted order, start in the sub-array, length}; gather = 1: sub[o + t] = order[s + t],
// else the way back
__global__ __launch_bounds__(kBlock) void run_copy_kernel(const int *__restrict__ table, int nruns, unsigned *__restrict__ order,
                                                          unsigned *__restrict__ sub, int gather)
{
    const int r = blockIdx.y;
    const int s0 = table[3 * r], o = table[3 * r + 1], len = table[3 * r + 2];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < len; t += gridDim.x * blockDim.x) {
        if (gather) sub[o + t] = order[s0 + t];
        else order[s0 + t] = sub[o + t];
    }
}

// head[j] = 1 when the j-th row in sorted order differs from the one before it
__global__ __launch_bounds__(kBlock) void head_kernel(const double *__restrict__ pts, i64 n, int dim,
                                                      const unsigned *__restrict__ order, int *__restrict__ head)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int h = 1;
    if (j > 0) {
        const double *a = pts + (i64)order[j] * dim;
        const double *b = pts + (i64)order[j - 1] * dim;
        h = 0;
        for (int c = 0; c < dim; ++c) h |= a[c] != b[c] ? 1 : 0;
    }
    head[j] = h;
}

// before[j] = number of heads in front of row j, so a row's unique index is before[j] + head[j] - 1
__global__ __launch_bounds__(kBlock) void emit_kernel(const double *__restrict__ pts, i64 n, int dim,
                                                      const unsigned *__restrict__ order,
                                                      const int *__restrict__ head, const int *__restrict__ before,
                                                      double *__restrict__ uniq, i64 *__restrict__ inverse)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const i64 row = order[j];
    const i64 u = (i64)before[j] + head[j] - 1;
    inverse[row] = u;
    if (head[j])
        for (int c = 0; c < dim; ++c) uniq[u * dim + c] = pts[row * dim + c];
}

}  // namespace

// The sort above for other callers (the kNN tree's Morton keys): stable LSD passes over the key bits [first_shift,
// end_shift), 8 at a time, ping-pong between (ka, va) and (kb, vb); *in_a = the result lies in (ka, va).  scratch:
// mm_radix_sort_scratch(n) bytes.
size_t mm_radix_sort_scratch(i64 n)
{
    const i64 tiles = (n + kTile - 1) / kTile;
    const i64 ncounts = (i64)kBins * tiles;
    const i64 count_tiles = (ncounts + 1 + kScanTileItems - 1) / kScanTileItems;
    return 2 * mm_round256((size_t)(ncounts + 1) * sizeof(int)) + mm_round256((size_t)count_tiles * sizeof(int));
}

int mm_radix_sort_pairs(mm_context *ctx, unsigned long long *ka, unsigned long long *kb, unsigned *va, unsigned *vb, i64 n,
                        int first_shift, int end_shift, void *scratch, bool *in_a)
{
    *in_a = true;
    if (n <= 0) return MM_OK;
    const int tiles = (int)((n + kTile - 1) / kTile);
    const i64 ncounts = (i64)kBins * tiles;
    char *sp = (char *)scratch;
    int *counts = (int *)sp;
    sp += mm_round256((size_t)(ncounts + 1) * sizeof(int));
    int *offsets = (int *)sp;
    sp += mm_round256((size_t)(ncounts + 1) * sizeof(int));
    int *count_sums = (int *)sp;
    for (int shift = first_shift; shift < end_shift; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)tiles), dim3(kSortBlock), 0, ctx->stream, ka, n, shift, tiles, counts);
        const int src = mm_exclusive_scan_int(ctx, counts, ncounts, offsets, count_sums);
        if (src != MM_OK) return src;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)tiles), dim3(kSortBlock), 0, ctx->stream, ka, va, n, shift, tiles,
                           offsets, kb, vb);
        u64 *tk = ka;
        ka = kb;
        kb = tk;
        unsigned *tv = va;
        va = vb;
        vb = tv;
        *in_a = !*in_a;
    }
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int64_t mm_unique_points(mm_context *ctx, const double *points_d, int64_t npoints, int64_t dim,
                                    double *unique_d, int64_t *inverse_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(dim >= 1 && dim <= 3, "dim must be 1, 2 or 3");
    MM_REQUIRE(npoints >= 0 && npoints < (int64_t)0x7fffffff, "npoints out of range");
    if (npoints == 0) return 0;
    MM_REQUIRE(points_d && unique_d && inverse_d, "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    const i64 n = npoints;
    const size_t n_sz = (size_t)n;

    const int ntiles = (int)((n + 1 + kScanTileItems - 1) / kScanTileItems);
    const int sort_tiles = (int)((n + kTile - 1) / kTile);
    const i64 ncounts = (i64)kBins * sort_tiles;
    const int count_tiles = (int)((ncounts + 1 + kScanTileItems - 1) / kScanTileItems);
    const size_t need = 2 * mm_round256(n_sz * sizeof(u64)) + 2 * mm_round256(n_sz * sizeof(unsigned)) +
                        2 * mm_round256((n_sz + 1) * sizeof(int)) + mm_round256((size_t)ntiles * sizeof(int)) +
                        2 * mm_round256((size_t)(ncounts + 1) * sizeof(int)) + mm_round256((size_t)count_tiles * sizeof(int)) +
                        2 * mm_round256((n_sz / 4 + 1) * sizeof(unsigned)) + mm_round256(sizeof(int) * (1 + 5 * (size_t)kMaxLongRuns)) +
                        4096;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) return rc;
    u64 *key_a = (u64 *)mm_scratch_take(ctx, n_sz * sizeof(u64));
    u64 *key_b = (u64 *)mm_scratch_take(ctx, n_sz * sizeof(u64));
    unsigned *ord_a = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    unsigned *ord_b = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    int *head = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *before = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
    int *counts = (int *)mm_scratch_take(ctx, (size_t)(ncounts + 1) * sizeof(int));
    int *offsets = (int *)mm_scratch_take(ctx, (size_t)(ncounts + 1) * sizeof(int));
    int *count_sums = (int *)mm_scratch_take(ctx, (size_t)count_tiles * sizeof(int));
    int *long_runs = (int *)mm_scratch_take(ctx, sizeof(int) * (1 + 5 * (size_t)kMaxLongRuns));   // count, starts, lengths, table
    unsigned *sub_a = (unsigned *)mm_scratch_take(ctx, (n_sz / 4 + 1) * sizeof(unsigned));
    unsigned *sub_b = (unsigned *)mm_scratch_take(ctx, (n_sz / 4 + 1) * sizeof(unsigned));
    if (!key_a || !key_b || !ord_a || !ord_b || !head || !before || !tile_sums || !counts || !offsets || !count_sums ||
        !long_runs || !sub_a || !sub_b) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        return MM_ERR_ALLOC;
    }

    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    // stable sort of (ka, va)[0 .. cnt) by the keys' 64 bits with (kb, vb) as the other buffer; eight passes, so the
    // result is back in (ka, va)
    // (first_shift: bits below it are not sorted by; the number of passes stays even)
    auto radix_sort = [&](u64 *ka, u64 *kb, unsigned *va, unsigned *vb, i64 cnt, int first_shift) -> int {
        const int tiles = (int)((cnt + kTile - 1) / kTile);
        for (int shift = first_shift; shift < 64; shift += 8) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)tiles), dim3(kSortBlock), 0, ctx->stream, ka, cnt, shift, tiles,
                               counts);
            int src = mm_exclusive_scan_int(ctx, counts, (i64)kBins * tiles, offsets, count_sums);
            if (src != MM_OK) return src;
            hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)tiles), dim3(kSortBlock), 0, ctx->stream, ka, va, cnt, shift,
                               tiles, offsets, kb, vb);
            u64 *tk = ka;
            ka = kb;
            kb = tk;
            unsigned *tv = va;
            va = vb;
            vb = tv;
        }
        return MM_OK;
    };
    // rows (va, any order that is ascending in the original index among equal rows) -> lexicographic order: dim stable
    // sorts, least significant coordinate first
    auto sort_rows = [&](u64 *ka, u64 *kb, unsigned *va, unsigned *vb, i64 cnt) -> int {
        const dim3 g((unsigned)((cnt + kBlock - 1) / kBlock));
        for (int comp = (int)dim - 1; comp >= 0; --comp) {
            hipLaunchKernelGGL(key_kernel, g, block, 0, ctx->stream, points_d, cnt, (int)dim, comp, va, ka);
            int src = radix_sort(ka, kb, va, vb, cnt, 0);
            if (src != MM_OK) return src;
        }
        return MM_OK;
    };
    // fast path: one sort by x, then the runs of equal x: short ones in place, long ones as a sub-sort
    static const bool force_general = getenv("MM_UNIQUE_GENERAL") != nullptr;   // (tests: the dim-sorts path on any input)
    bool general = force_general;
    if (!general) {
        hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, ord_a, n);
        hipLaunchKernelGGL(key_kernel, grid, block, 0, ctx->stream, points_d, n, (int)dim, 0, ord_a, key_a);
        static_assert(kKeyLowBits % 16 == 0, "an even number of 8-bit passes");
        // (1-D: nothing follows that could order the rest of x -- all 64 bits)
        if ((rc = radix_sort(key_a, key_b, ord_a, ord_b, n, dim > 1 ? kKeyLowBits : 0)) != MM_OK) return rc;
        if (dim > 1) {
            MM_HIP_CHECK(hipMemsetAsync(long_runs, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(run_fixup_kernel, grid, block, 0, ctx->stream, key_a, n, points_d, (int)dim, ord_a, long_runs);
            int nlong = 0;
            MM_HIP_CHECK(hipMemcpyAsync(&nlong, long_runs, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (nlong > kMaxLongRuns) {
                general = true;
            } else if (nlong > 0) {
                // lengths by bisection on the device, the layout of the sub-array on the host (a few runs)
                int *d_start = long_runs + 1, *d_len = long_runs + 1 + kMaxLongRuns, *d_table = long_runs + 1 + 2 * kMaxLongRuns;
                hipLaunchKernelGGL(run_length_kernel, dim3((unsigned)((nlong + kBlock - 1) / kBlock)), block, 0, ctx->stream, key_a,
                                   n, d_start, nlong, d_len);
                std::vector<int> h(2 * (size_t)kMaxLongRuns);
                MM_HIP_CHECK(hipMemcpyAsync(h.data(), d_start, sizeof(int) * (size_t)nlong, hipMemcpyDeviceToHost, ctx->stream));
                MM_HIP_CHECK(hipMemcpyAsync(h.data() + kMaxLongRuns, d_len, sizeof(int) * (size_t)nlong, hipMemcpyDeviceToHost,
                                            ctx->stream));
                MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                std::vector<int> by_start((size_t)nlong);
                for (int r = 0; r < nlong; ++r) by_start[(size_t)r] = r;
                std::sort(by_start.begin(), by_start.end(), [&](int a, int b) { return h[(size_t)a] < h[(size_t)b]; });
                std::vector<int> table(3 * (size_t)nlong);
                i64 m = 0;
                int longest = 0;
                for (int q = 0; q < nlong; ++q) {
                    const int r = by_start[(size_t)q], len = h[(size_t)kMaxLongRuns + r];
                    table[3 * (size_t)q] = h[(size_t)r];
                    table[3 * (size_t)q + 1] = (int)m;
                    table[3 * (size_t)q + 2] = len;
                    m += len;
                    longest = len > longest ? len : longest;
                }
                if (m > n / 4) {
                    general = true;   // a structured cloud: most rows sit in long runs
                } else {
                    MM_HIP_CHECK(hipMemcpyAsync(d_table, table.data(), sizeof(int) * 3 * (size_t)nlong, hipMemcpyHostToDevice,
                                                ctx->stream));
                    // the runs' rows side by side in x order -> sorted by (x, y, z, index) -> back to the runs' places: both
                    // orders are x-major, so sub-array position o_r + t IS place start_r + t of the result
                    int gx = (longest + kBlock - 1) / kBlock;
                    if (gx > 1024) gx = 1024;
                    hipLaunchKernelGGL(run_copy_kernel, dim3((unsigned)gx, (unsigned)nlong), block, 0, ctx->stream, d_table, nlong,
                                       ord_a, sub_a, 1);
                    if ((rc = sort_rows(key_b, key_b + m, sub_a, sub_b, m)) != MM_OK) return rc;
                    hipLaunchKernelGGL(run_copy_kernel, dim3((unsigned)gx, (unsigned)nlong), block, 0, ctx->stream, d_table, nlong,
                                       ord_a, sub_a, 0);
                    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // (the host table goes out of scope)
                }
            }
        }
    }
    if (general) {
        // long runs of equal x everywhere (a structured cloud)
        hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, ord_a, n);
        if ((rc = sort_rows(key_a, key_b, ord_a, ord_b, n)) != MM_OK) return rc;
    }
    hipLaunchKernelGGL(head_kernel, grid, block, 0, ctx->stream, points_d, n, (int)dim, ord_a, head);
    rc = mm_exclusive_scan_int(ctx, head, n, before, tile_sums);   // before[n] = number of unique rows
    if (rc != MM_OK) return rc;
    hipLaunchKernelGGL(emit_kernel, grid, block, 0, ctx->stream, points_d, n, (int)dim, ord_a, head, before, unique_d,
                       (i64 *)inverse_d);
    MM_HIP_CHECK(hipGetLastError());
    int nunique = 0;
    MM_HIP_CHECK(hipMemcpyAsync(&nunique, before + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return (int64_t)nunique;
}
