// Device passes of the layer-aware GLL drivers.  The reference loops over the
// layers of an Earth model on the host (components/interpolator.py:1047-1082): per layer a mask of source
// elements, a tree over their centroids, get_element_weights for the unique points of the target elements
// of that layer, then per parameter  new_field[mask[layer]] = values[inverse].reshape(...)  -- and, in the
// un-layered gll_2_gll, a fluid/solid fix-up of the interpolated element data (:829-841).  The per-layer
// interpolation itself is mm_interpolate_gll on the layer's sub-meshes; what lives here are the two passes
// around it that would otherwise pull the element-nodal arrays through the host:
//   mm_scatter_elements  : values[inverse] written back into the rows of the layer's target elements
//   mm_fluid_solid_fix   : interpolator.py:829-841 on the device
#include "mm_common.h"

namespace {

// out[c][elem_ids[m]][p] = values[inverse[m * P + p]][c]   (values point-major [U][C], out [C][nelem_out][P])
__global__ __launch_bounds__(256) void scatter_elements_kernel(const double *__restrict__ values, i64 nunique, int ncomp,
                                                               const i64 *__restrict__ inverse,
                                                               const i64 *__restrict__ elem_ids, i64 nmasked, int P,
                                                               i64 nelem_out, double *__restrict__ out)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nmasked * P) return;
    const i64 m = t / P;
    const int p = (int)(t - m * P);
    const i64 e = elem_ids[m];
    const i64 u = inverse[t];
    if ((unsigned long long)e >= (unsigned long long)nelem_out || (unsigned long long)u >= (unsigned long long)nunique) return;
    for (int c = 0; c < ncomp; ++c) out[((i64)c * nelem_out + e) * P + p] = values[u * ncomp + c];
}

// One wave per target element: a fluid element keeps its previous values; a solid element keeps them too
// when the interpolated shear velocity is exactly zero at any of its points (fluid values that leaked
// into the solid part).  values / previous element-major [nelem][ncomp][P].
__global__ __launch_bounds__(64) void fluid_solid_fix_kernel(double *__restrict__ values,
                                                             const double *__restrict__ previous,
                                                             const unsigned char *__restrict__ solid, i64 nelem,
                                                             int ncomp, int P, int vs_index,
                                                             unsigned long long *__restrict__ restored)
{
    const i64 e = blockIdx.x;
    const int lane = threadIdx.x;
    const bool is_solid = solid[e] != 0;
    bool zero = false;
    if (is_solid)
        for (int p = lane; p < P; p += 64) zero = zero || values[(e * ncomp + vs_index) * P + p] == 0.0;
    const bool restore = !is_solid || __any(zero);
    if (!restore) return;
    for (i64 q = lane; q < (i64)ncomp * P; q += 64) values[e * ncomp * P + q] = previous[e * ncomp * P + q];
    if (lane == 0 && is_solid) atomicAdd(restored, 1ull);
}

// index of a GLL point -> index of its element: floor(index / P) (reference interpolator.py:113, :777), in place
__global__ __launch_bounds__(256) void points_to_elements_kernel(i64 *__restrict__ idx, i64 n, i64 P)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const i64 v = idx[i];
        idx[i] = v >= 0 ? v / P : -((-v + P - 1) / P);   // floor, like np.floor(idx / P)
    }
}

}  // namespace

extern "C" int mm_points_to_elements(mm_context *ctx, int64_t *idx_d, int64_t n, int64_t P)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(n >= 0 && P >= 1, "bad size");
    if (n == 0) return MM_OK;
    MM_REQUIRE(idx_d != nullptr, "null array");
    MM_REQUIRE((n + 255) / 256 < (i64)0x7fffffff, "too many indices for one launch");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(points_to_elements_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (i64 *)idx_d, n, P);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int mm_scatter_elements(mm_context *ctx, const double *values_d, int64_t nunique, int64_t ncomp,
                                   const int64_t *inverse_d, const int64_t *elem_ids_d, int64_t nmasked, int64_t P,
                                   int64_t nelem_out, double *out_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(nunique >= 0 && ncomp >= 0 && nmasked >= 0 && nelem_out >= 0, "negative size");
    MM_REQUIRE(P >= 1 && P <= 4096 && ncomp < (1 << 20), "P or ncomp out of range");
    if (nmasked == 0 || ncomp == 0) return MM_OK;
    MM_REQUIRE(values_d && inverse_d && elem_ids_d && out_d, "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    const i64 total = nmasked * P;
    MM_REQUIRE((total + 255) / 256 < (i64)0x7fffffff, "too many points for one launch");
    hipLaunchKernelGGL(scatter_elements_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, values_d,
                       nunique, (int)ncomp, (const i64 *)inverse_d, (const i64 *)elem_ids_d, nmasked, (int)P, nelem_out, out_d);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int64_t mm_fluid_solid_fix(mm_context *ctx, double *values_d, const double *previous_d,
                                      const unsigned char *solid_d, int64_t nelem, int64_t ncomp, int64_t P,
                                      int64_t vs_index)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(nelem >= 0 && nelem < (int64_t)0x7fffffff, "nelem out of range");
    MM_REQUIRE(ncomp >= 1 && ncomp < (1 << 20) && P >= 1 && P <= 4096, "ncomp or P out of range");
    MM_REQUIRE(vs_index >= 0 && vs_index < ncomp, "vs_index out of range");
    if (nelem == 0) return 0;
    MM_REQUIRE(values_d && previous_d && solid_d, "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipMemsetAsync(ctx->d_counters + 1, 0, sizeof(i64), ctx->stream));
    hipLaunchKernelGGL(fluid_solid_fix_kernel, dim3((unsigned)nelem), dim3(64), 0, ctx->stream, values_d, previous_d,
                       solid_d, nelem, (int)ncomp, (int)P, (int)vs_index, (unsigned long long *)(ctx->d_counters + 1));
    MM_HIP_CHECK(hipGetLastError());
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 1, ctx->d_counters + 1, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ctx->h_counters[1];
}
