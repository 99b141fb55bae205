// A1 -- element centroids.  Replaces reference multi_mesh/src/centroid.c:3-25.
//
// One lane per element.  Per axis the element's node coordinates are summed in connectivity
// order starting from 0.0 and the sum is DIVIDED by the node count (centroid.c:17-22), so the
// result is bit-identical to the reference.  HBM-bound: E * (P*8 id bytes + P*ndim*8 gathered
// coordinate bytes + ndim*8 written); the coordinate gathers hit L2 for a mesh-ordered
// connectivity.
#include "mm_common.h"

template <int NDIM, int NPER>
__global__ __launch_bounds__(256) void centroid_kernel(i64 nelem, i64 nper_rt,
                                                       const i64 *__restrict__ conn,
                                                       const double *__restrict__ points,
                                                       double *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nelem) return;
    const i64 nper = NPER > 0 ? NPER : nper_rt;
    const i64 *row = conn + e * nper;
    double acc[NDIM];
#pragma unroll
    for (int a = 0; a < NDIM; ++a) acc[a] = 0.;
    if (NPER > 0) {
        i64 id[NPER > 0 ? NPER : 1];
#pragma unroll
        for (int p = 0; p < NPER; ++p) id[p] = row[p];
#pragma unroll
        for (int p = 0; p < NPER; ++p) {
#pragma unroll
            for (int a = 0; a < NDIM; ++a) acc[a] = acc[a] + points[id[p] * NDIM + a];
        }
    } else {
        for (i64 p = 0; p < nper; ++p) {
            const i64 id = row[p];
#pragma unroll
            for (int a = 0; a < NDIM; ++a) acc[a] = acc[a] + points[id * NDIM + a];
        }
    }
    const double denom = (double)nper;
#pragma unroll
    for (int a = 0; a < NDIM; ++a) out[e * NDIM + a] = acc[a] / denom;
}

// Fused-pipeline variant (hex8, 3-D): the same centroids, and each workgroup also leaves the bounding
// box of the centroids it produced in partial[block][6] (min x,y,z, max x,y,z) -- the search grid of
// the next stage is sized from that box, and taking it here saves a pass over the centroid array.
// Grid-stride over the elements so that the number of partials stays small.
__global__ __launch_bounds__(256) void centroid_bbox_kernel(i64 nelem, const i64 *__restrict__ conn,
                                                            const double *__restrict__ points,
                                                            double *__restrict__ out, double *__restrict__ partial)
{
    __shared__ double s_box[6][256 / 64];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // (the connectivity row of a thread's NEXT element is requested before the node coordinates of the current one
    // are gathered: one dependent round trip per element instead of two)
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 idn[8];
    {
        const i64 *row = conn + (e < nelem ? e : 0) * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) idn[p] = row[p];
    }
    for (; e < nelem; e += stride) {
        i64 id[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) id[p] = idn[p];
        {
            const i64 *row = conn + (e + stride < nelem ? e + stride : e) * 8;
#pragma unroll
            for (int p = 0; p < 8; ++p) idn[p] = row[p];
        }
        double acc[3] = {0., 0., 0.};
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = acc[a] + points[id[p] * 3 + a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double c = acc[a] / 8.0;   // summed in connectivity order from 0.0, divided (centroid.c:17-22)
            out[e * 3 + a] = c;
            mn[a] = fmin(mn[a], c);
            mx[a] = fmax(mx[a], c);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fmin(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmax(mx[a], __shfl_xor(mx[a], off));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_box[a][wave] = mn[a];
            s_box[3 + a][wave] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = s_box[threadIdx.x][0];
        for (int w = 1; w < 256 / 64; ++w) v = threadIdx.x < 3 ? fmin(v, s_box[threadIdx.x][w]) : fmax(v, s_box[threadIdx.x][w]);
        partial[(i64)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

int mm_launch_centroid_bbox(mm_context *ctx, i64 nelem, const i64 *conn, const double *points, double *out,
                            double *partial, int nblocks)
{
    if (nelem == 0) return MM_OK;
    hipLaunchKernelGGL(centroid_bbox_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, nelem, conn, points,
                       out, partial);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

int mm_launch_centroid(mm_context *ctx, i64 ndim, i64 nelem, i64 nper, const i64 *conn,
                       const double *points, double *out)
{
    if (nelem == 0) return MM_OK;
    const int block = 256;
    const i64 grid = (nelem + block - 1) / block;
    MM_REQUIRE(grid < (i64)0x7fffffff, "too many elements for one launch");
    dim3 g((unsigned)grid), b(block);
    if (ndim == 3 && nper == 8)
        hipLaunchKernelGGL((centroid_kernel<3, 8>), g, b, 0, ctx->stream, nelem, nper, conn, points, out);
    else if (ndim == 3)
        hipLaunchKernelGGL((centroid_kernel<3, 0>), g, b, 0, ctx->stream, nelem, nper, conn, points, out);
    else if (ndim == 2 && nper == 4)
        hipLaunchKernelGGL((centroid_kernel<2, 4>), g, b, 0, ctx->stream, nelem, nper, conn, points, out);
    else if (ndim == 2)
        hipLaunchKernelGGL((centroid_kernel<2, 0>), g, b, 0, ctx->stream, nelem, nper, conn, points, out);
    else
        hipLaunchKernelGGL((centroid_kernel<1, 0>), g, b, 0, ctx->stream, nelem, nper, conn, points, out);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int mm_centroid(mm_context *ctx, int64_t ndim, int64_t nelem, int64_t nper,
                           const int64_t *conn_d, const double *points_d, double *centroid_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(ndim >= 1 && ndim <= 3, "ndim must be 1, 2 or 3");
    MM_REQUIRE(nelem >= 0 && nper >= 1, "bad sizes");
    MM_REQUIRE(nelem == 0 || (conn_d && points_d && centroid_d), "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    mm_stage_begin(ctx, MM_STAGE_CENTROID);
    int rc = mm_launch_centroid(ctx, ndim, nelem, nper, (const i64 *)conn_d, points_d, centroid_d);
    mm_stage_end(ctx, MM_STAGE_CENTROID);
    return rc;
}
