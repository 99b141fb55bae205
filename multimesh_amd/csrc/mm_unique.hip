// A11 -- unique target points with the index array that rebuilds the input.
// Replaces np.unique(points.reshape(-1, dim), axis=0, return_inverse=True) at reference
// multi_mesh/utils.py:484-488 (the pre-step of every GLL-target flow: element-nodal points repeat on
// shared faces, edges and corners, ~1/3 to 1/2 of them are duplicates; the scatter-back is
// interpolator.py:823).
//
// Rows are ordered lexicographically by (x, y, z) in fp64 comparison order, equal rows collapse to
// one, inverse[i] = position of row i in the unique list.  Done as dim stable least-significant-first
// sorts of (order-preserving 64-bit image of one coordinate, row index) pairs -- the radix sort is
// rocPRIM's (a plain library sort, like a library GEMM; everything specific to this path is below) --
// followed by a head-flag / prefix-sum / scatter pass.  -0.0 compares equal to +0.0 as in NumPy;
// the row kept for a group of equal rows is the one with the smallest original index (NumPy's
// unstable sort leaves that choice open when rows differ only in the sign of a zero).  NaN
// coordinates are not supported (they are ordered by bit pattern and never merge).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mm_common.h"

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums);

namespace {

constexpr int kBlock = 256;
constexpr int kScanTileItems = 1024;  // items per block of mm_exclusive_scan_int

// order-preserving map double -> uint64 (-0.0 first folded into +0.0)
__device__ __forceinline__ unsigned long long sortable(double v)
{
    if (v == 0.0) v = 0.0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(kBlock) void iota_kernel(unsigned *__restrict__ idx, i64 n)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (unsigned)i;
}

// key of coordinate `comp` of the rows in their current order
__global__ __launch_bounds__(kBlock) void key_kernel(const double *__restrict__ pts, i64 n, int dim, int comp,
                                                     const unsigned *__restrict__ order,
                                                     unsigned long long *__restrict__ key)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) key[j] = sortable(pts[(i64)order[j] * dim + comp]);
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

    size_t sort_bytes = 0;
    MM_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long *)nullptr,
                                           (unsigned long long *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr, n_sz,
                                           0, 64, ctx->stream));
    const int ntiles = (int)((n + 1 + kScanTileItems - 1) / kScanTileItems);
    const size_t need = 2 * mm_round256(n_sz * sizeof(unsigned long long)) + 2 * mm_round256(n_sz * sizeof(unsigned)) +
                        2 * mm_round256((n_sz + 1) * sizeof(int)) + mm_round256((size_t)ntiles * sizeof(int)) +
                        mm_round256(sort_bytes) + 4096;
    int rc = mm_scratch_begin(ctx, need);
    if (rc != MM_OK) return rc;
    unsigned long long *key_a = (unsigned long long *)mm_scratch_take(ctx, n_sz * sizeof(unsigned long long));
    unsigned long long *key_b = (unsigned long long *)mm_scratch_take(ctx, n_sz * sizeof(unsigned long long));
    unsigned *ord_a = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    unsigned *ord_b = (unsigned *)mm_scratch_take(ctx, n_sz * sizeof(unsigned));
    int *head = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *before = (int *)mm_scratch_take(ctx, (n_sz + 1) * sizeof(int));
    int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
    void *sort_tmp = mm_scratch_take(ctx, sort_bytes ? sort_bytes : 256);
    if (!key_a || !key_b || !ord_a || !ord_b || !head || !before || !tile_sums || !sort_tmp) {
        mm_set_error(MM_ERR_ALLOC, "scratch carve failed");
        return MM_ERR_ALLOC;
    }

    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, ord_a, n);
    // least significant coordinate first; every sort is stable, so after the last one the rows are in
    // lexicographic order and equal rows in order of their original index
    for (int comp = (int)dim - 1; comp >= 0; --comp) {
        hipLaunchKernelGGL(key_kernel, grid, block, 0, ctx->stream, points_d, n, (int)dim, comp, ord_a, key_a);
        MM_HIP_CHECK(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, key_a, key_b, ord_a, ord_b, n_sz, 0, 64, ctx->stream));
        unsigned *t = ord_a;
        ord_a = ord_b;
        ord_b = t;
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
