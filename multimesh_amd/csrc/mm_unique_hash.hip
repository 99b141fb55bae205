// A11, order-free form -- the unique rows of a point array and the index array that rebuilds the input, WITHOUT
// np.unique's lexicographic order: rows come out in the order of their first occurrence.
//
// np.unique(points, axis=0, return_inverse=True) (reference multi_mesh/utils.py:484-488) is only ever used by the
// reference to collapse the element-nodal target points before the interpolation and to scatter the values back through
// the inverse (components/interpolator.py:823, :1079-1081): the ORDER of the unique rows never reaches a result.
// mm_unique_points reproduces NumPy's order with six radix passes over 13 M keys plus fix-ups (3.2 ms for cfg5's
// target mesh: a third of that step); where the caller does not need the order -- the fused GLL drivers, bench.py's
// cfg5 step -- this form does the same job with a hash table: one insertion pass (a slot per distinct row, claimed with
// one compare-and-swap, keeping the SMALLEST row index of its class), one pass that marks the class representatives,
// a prefix sum, one pass that writes the unique rows and the inverse.  Deterministic: the representative of a class is
// its smallest index whatever the order of the atomics, and the output order is ascending representative index.
// -0.0 equals +0.0 as in NumPy (the row kept is the first one); NaN rows never merge.
#include "mm_common.h"

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums);

namespace {

typedef unsigned long long u64;
constexpr int kBlock = 256;
constexpr int kScanTileItems = 1024;

__device__ __forceinline__ u64 canon_bits(double v)
{
    if (v == 0.0) v = 0.0;   // -0.0 -> +0.0
    return (u64)__double_as_longlong(v);
}

__device__ __forceinline__ u64 mix64(u64 h)
{
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

template <int DIM>
__device__ __forceinline__ void load_row(const double *__restrict__ pts, i64 i, u64 (&b)[3])
{
    b[0] = canon_bits(pts[i * DIM]);
    b[1] = DIM > 1 ? canon_bits(pts[i * DIM + 1]) : 0ull;
    b[2] = DIM > 2 ? canon_bits(pts[i * DIM + 2]) : 0ull;
}

// slot_of[i] = the table slot of row i's class; table[slot] = smallest row index of the class seen so far
template <int DIM>
__global__ __launch_bounds__(kBlock) void unique_insert_kernel(const double *__restrict__ pts, i64 n, int *__restrict__ table,
                                                               unsigned mask, unsigned *__restrict__ slot_of)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 b[3];
    load_row<DIM>(pts, i, b);
    unsigned s = (unsigned)mix64(b[0] ^ mix64(b[1] ^ mix64(b[2] + 0x9e3779b97f4a7c15ull))) & mask;
    for (;;) {
        int cur = __atomic_load_n(&table[s], __ATOMIC_RELAXED);
        if (cur < 0) {
            const int old = atomicCAS(&table[s], -1, (int)i);
            if (old < 0) break;   // claimed
            cur = old;
        }
        u64 c[3];
        load_row<DIM>(pts, (i64)cur, c);
        if (c[0] == b[0] && c[1] == b[1] && c[2] == b[2]) {
            if ((int)i < cur) atomicMin(&table[s], (int)i);
            break;
        }
        s = (s + 1) & mask;
    }
    slot_of[i] = s;
}

__global__ __launch_bounds__(kBlock) void unique_flag_kernel(const int *__restrict__ table, const unsigned *__restrict__ slot_of,
                                                             i64 n, int *__restrict__ rep, int *__restrict__ flag)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = table[slot_of[i]];
    rep[i] = r;
    flag[i] = r == (int)i ? 1 : 0;
}

template <int DIM>
__global__ __launch_bounds__(kBlock) void unique_emit_kernel(const double *__restrict__ pts, i64 n, const int *__restrict__ rep,
                                                             const int *__restrict__ rank, double *__restrict__ unique,
                                                             i64 *__restrict__ inverse)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rep[i];
    const int at = rank[r];
    inverse[i] = (i64)at;
    if (r == (int)i) {
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            double v = pts[i * DIM + a];
            if (v == 0.0) v = 0.0;   // NumPy's unique row of a {-0.0, +0.0} class is whichever sorts first; here: +0.0
            unique[(i64)at * DIM + a] = v;
        }
    }
}

template <int DIM>
int64_t unique_any_order(mm_context *ctx, const double *pts, i64 n, double *unique_d, i64 *inverse_d)
{
    unsigned tsize = 1024;
    while ((u64)tsize < 2ull * (u64)n) tsize <<= 1;
    const size_t n_sz = (size_t)n;
    const int ntiles = (int)((n + 1 + kScanTileItems - 1) / kScanTileItems);
    const size_t need = mm_round256((size_t)tsize * sizeof(int)) + mm_round256(n_sz * sizeof(unsigned)) + mm_round256(n_sz * sizeof(int)) +
                        2 * mm_round256((n_sz + 1) * sizeof(int)) + mm_round256((size_t)ntiles * sizeof(int)) + 2048;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) return rc;
    int *table = (int *)mm_scratch_take(ctx, (size_t)tsize * sizeof(int));
    unsigned *slot_of = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    int *rep = (int *)mm_scratch_take(ctx, n_sz * sizeof(int));
    int *flag = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *rank = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
    if (!table || !slot_of || !rep || !flag || !rank || !tile_sums) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        return MM_ERR_ALLOC;
    }
    MM_HIP_CHECK(hipMemsetAsync(table, 0xff, (size_t)tsize * sizeof(int), ctx->stream));
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL((unique_insert_kernel<DIM>), grid, block, 0, ctx->stream, pts, n, table, tsize - 1u, slot_of);
    hipLaunchKernelGGL(unique_flag_kernel, grid, block, 0, ctx->stream, table, slot_of, n, rep, flag);
    MM_HIP_CHECK(hipGetLastError());
    rc = mm_exclusive_scan_int(ctx, flag, n, rank, tile_sums);   // rank[0 .. n), rank[n] = the number of unique rows
    if (rc != MM_OK) return rc;
    hipLaunchKernelGGL((unique_emit_kernel<DIM>), grid, block, 0, ctx->stream, pts, n, rep, rank, unique_d, inverse_d);
    MM_HIP_CHECK(hipGetLastError());
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 2, rank + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return (int64_t) * reinterpret_cast<const int *>(ctx->h_counters + 2);
}

}  // namespace

extern "C" int64_t mm_unique_points_any_order(mm_context *ctx, const double *points_d, int64_t npoints, int64_t dim,
                                              double *unique_d, int64_t *inverse_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(dim >= 1 && dim <= 3, "dim must be 1, 2 or 3");
    MM_REQUIRE(npoints >= 0 && npoints < (int64_t)0x3fffffff, "npoints out of range");
    if (npoints == 0) return 0;
    MM_REQUIRE(points_d && unique_d && inverse_d, "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    if (dim == 3) return unique_any_order<3>(ctx, points_d, npoints, unique_d, (i64 *)inverse_d);
    if (dim == 2) return unique_any_order<2>(ctx, points_d, npoints, unique_d, (i64 *)inverse_d);
    return unique_any_order<1>(ctx, points_d, npoints, unique_d, (i64 *)inverse_d);
}
