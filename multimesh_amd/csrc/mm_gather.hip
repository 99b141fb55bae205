// A9 -- shape-function weighted gather.  Replaces np.sum(field[ids] * w, axis=1) at reference
// multi_mesh/scripts/cli.py:98-100 (hex8, P = 8) and components/interpolator.py:976 (P = 27, 125).
//
// Bit parity with NumPy: each product field[id]*w is rounded on its own (no fused multiply-add,
// the file is built with -ffp-contract=off) and the row is summed in NumPy's pairwise add-reduce
// order for a contiguous row of P < 128 doubles: eight running partials r[j] += a[8i+j], folded as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the P%8 tail added in order (P < 8: sequential from 0).
// The fold is a lane butterfly: floating-point addition is commutative, so the xor-1 / xor-2 /
// mirror exchange produces exactly those parenthesised sums in every lane.
//
// HBM-bound.  Algorithmic bytes per target: P*8 (ids) + P*8 (weights) + C*(P*8 gathered + 8 out)
// = 128 + 72*C for hex8.  ids/weights are the streaming part and are read with 16-byte
// (hex8) / 8-byte (general) fully coalesced lane accesses; the gathered field reads are 8-byte
// random accesses served mostly by L2 / Infinity Cache when targets arrive in a spatially
// coherent order.
#include "mm_common.h"

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v)
{
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each group of 8

// A node id outside [0, nsrc) would fault the GPU (the reference would read out of bounds on
// the host); such ids read node 0 instead.
__device__ __forceinline__ i64 safe_id(i64 id, i64 nsrc)
{
    return (unsigned long long)id < (unsigned long long)nsrc ? id : 0;
}

// ---- hex8: 4 lanes per target, each lane owns entries 2q and 2q+1 (one 16-byte load each of
// ids and weights), 16 targets per wave-instruction.
template <bool POINT_MAJOR>
__global__ __launch_bounds__(256) void gather8_kernel(const double *__restrict__ fields, i64 nsrc,
                                                      int ncomp, const longlong2 *__restrict__ ids,
                                                      const double2 *__restrict__ w, i64 npoints,
                                                      double *__restrict__ out)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n_raw = t >> 2;
    const int q = (int)(t & 3);
    const bool valid = n_raw < npoints;
    const i64 n = valid ? n_raw : npoints - 1;  // keep every lane live for the lane exchanges
    const longlong2 id2 = ids[n * 4 + q];
    const double2 w2 = w[n * 4 + q];
    for (int c = 0; c < ncomp; ++c) {
        const double *f = fields + (i64)c * nsrc;
        const double a0 = f[safe_id(id2.x, nsrc)] * w2.x;
        const double a1 = f[safe_id(id2.y, nsrc)] * w2.y;
        double s = a0 + a1;
        s = s + dpp_move_f64<DPP_QUAD_XOR1>(s);
        s = s + dpp_move_f64<DPP_QUAD_XOR2>(s);
        if (valid && q == 0) {
            // NumPy starts a reduction from the identity: 0.0 + (row sum), visible where the sum is -0.0
            if (POINT_MAJOR) out[n * ncomp + c] = 0.0 + s;
            else out[(i64)c * npoints + n] = 0.0 + s;
        }
    }
}

// ---- general P (27, 125, 25, 4 ...): 8 lanes per target; lane j owns the running partial r[j].
template <bool POINT_MAJOR>
__global__ __launch_bounds__(256) void gatherP_kernel(const double *__restrict__ fields, i64 nsrc,
                                                      int ncomp, const i64 *__restrict__ ids,
                                                      const double *__restrict__ w, i64 npoints,
                                                      int P, double *__restrict__ out)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n_raw = t >> 3;
    const int j = (int)(t & 7);
    const int lane = threadIdx.x & 63;
    const int group_base = lane & ~7;
    const bool valid = n_raw < npoints;
    const i64 n = valid ? n_raw : npoints - 1;
    const i64 *idrow = ids + n * P;
    const double *wrow = w + n * P;
    const int tail = P & 7;
    const int nfull = P - tail;
    for (int c = 0; c < ncomp; ++c) {
        const double *f = fields + (i64)c * nsrc;
        double res;
        if (P < 8) {
            const double a = j < P ? f[safe_id(idrow[j], nsrc)] * wrow[j] : 0.0;
            res = 0.;
            for (int i = 0; i < P; ++i) res += __shfl(a, group_base + i);
        } else {
            double r = f[safe_id(idrow[j], nsrc)] * wrow[j];
            for (int i = 8; i < nfull; i += 8) r += f[safe_id(idrow[i + j], nsrc)] * wrow[i + j];
            double s = r + dpp_move_f64<DPP_QUAD_XOR1>(r);
            s = s + dpp_move_f64<DPP_QUAD_XOR2>(s);
            s = s + dpp_move_f64<DPP_HALF_MIRROR>(s);
            const double a = j < tail ? f[safe_id(idrow[nfull + j], nsrc)] * wrow[nfull + j] : 0.0;
            res = s;
            for (int i = 0; i < tail; ++i) res += __shfl(a, group_base + i);
        }
        if (valid && j == 0) {
            if (POINT_MAJOR) out[n * ncomp + c] = 0.0 + res;   // from the identity, as above
            else out[(i64)c * npoints + n] = 0.0 + res;
        }
    }
}

int mm_launch_gather(mm_context *ctx, const double *fields, i64 nsrc, i64 ncomp, const i64 *ids,
                     const double *w, i64 npoints, i64 P, double *out, int out_point_major)
{
    if (npoints == 0 || ncomp == 0) return MM_OK;
    const int block = 256;
    const bool fast8 = (P == 8) && (((uintptr_t)ids & 15) == 0) && (((uintptr_t)w & 15) == 0);
    const i64 lanes_per_target = fast8 ? 4 : 8;
    const i64 grid = (npoints * lanes_per_target + block - 1) / block;
    MM_REQUIRE(grid < (i64)0x7fffffff, "too many targets for one launch");
    dim3 g((unsigned)grid), b(block);
    if (fast8) {
        if (out_point_major)
            hipLaunchKernelGGL((gather8_kernel<true>), g, b, 0, ctx->stream, fields, nsrc, (int)ncomp,
                               (const longlong2 *)ids, (const double2 *)w, npoints, out);
        else
            hipLaunchKernelGGL((gather8_kernel<false>), g, b, 0, ctx->stream, fields, nsrc, (int)ncomp,
                               (const longlong2 *)ids, (const double2 *)w, npoints, out);
    } else {
        if (out_point_major)
            hipLaunchKernelGGL((gatherP_kernel<true>), g, b, 0, ctx->stream, fields, nsrc, (int)ncomp,
                               ids, w, npoints, (int)P, out);
        else
            hipLaunchKernelGGL((gatherP_kernel<false>), g, b, 0, ctx->stream, fields, nsrc, (int)ncomp,
                               ids, w, npoints, (int)P, out);
    }
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int mm_gather(mm_context *ctx, const double *fields_d, int64_t nsrc, int64_t ncomp,
                         const int64_t *ids_d, const double *w_d, int64_t npoints, int64_t P,
                         double *out_d, int out_point_major)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(nsrc >= 0 && ncomp >= 0 && npoints >= 0, "negative size");
    MM_REQUIRE(P >= 1 && P <= 128, "P must be in 1..128");
    MM_REQUIRE(ncomp < (1 << 20), "ncomp too large");
    MM_REQUIRE(npoints == 0 || ncomp == 0 || (fields_d && ids_d && w_d && out_d), "null array");
    MM_REQUIRE(npoints == 0 || ncomp == 0 || nsrc >= 1, "empty source field");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    mm_stage_begin(ctx, MM_STAGE_GATHER);
    int rc = mm_launch_gather(ctx, fields_d, nsrc, ncomp, (const i64 *)ids_d, w_d, npoints, P, out_d,
                              out_point_major);
    mm_stage_end(ctx, MM_STAGE_GATHER);
    return rc;
}
