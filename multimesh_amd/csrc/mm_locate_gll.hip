// A10 -- GLL elements (order 1, 2, 4; 2-D and 3-D): element search by Newton inversion of the
// isoparametric map, tensor-product Lagrange interpolation coefficients, element-nodal gather.
// Replaces the per-point Python loops of reference multi_mesh/components/interpolator.py:
//   get_element_weights.check_inside :1181-1233 (control flow reproduced here),
//   inverse_transform :1370-1386 and get_coefficients :1337-1347 (thin wrappers over the
//   proprietary salvus.fem package, which is absent: PARITY UNPINNED -- the numerics below are
//   this project's own definition, identical to oracle/mm_oracle.c "A10", pinned by analytic
//   properties: partition of unity, Kronecker property at the nodes, exact reproduction of
//   polynomials up to the element order, order-1 == the hex8 path),
//   and the gather np.sum(coeffs * field[elem_indices], axis=1) :976.
//
// Numerics: tensor-product Lagrange basis on the GLL nodes (order 1: -1,1; order 2: -1,0,1;
// order 4: -1,-sqrt(3/7),0,sqrt(3/7),1), node index p = i + (n+1) j + (n+1)^2 k; Newton from
// xi = 0 with the analytic Jacobian and a cofactor solve; converged when the largest update
// component is < 1e-10, at most 25 updates; NaN when an iterate leaves [-10,10] or the iteration
// does not converge.  Same operation order as the oracle, no fused multiply-add.
//
// One lane per target, control nodes streamed from L1/L2 each Newton step.  Targets are visited in
// the order of their FIRST candidate element (a counting sort by nn[:,0] on the device), so the
// lanes of a wave mostly work on the same element and their control-node loads collapse to one
// cache line per instruction instead of up to 64.  (An element-centric LDS-tiled variant is the
// next step, see DESIGN.md.)
#include <math.h>

#include "mm_common.h"
#include "mm_newton_hex8.h"

namespace {

template <int ORDER>
__device__ __forceinline__ void gll_nodes(double (&g)[ORDER + 1])
{
    if (ORDER == 1) {
        g[0] = -1.0;
        g[ORDER] = 1.0;
    } else if (ORDER == 2) {
        g[0] = -1.0;
        g[1] = 0.0;
        g[ORDER] = 1.0;
    } else {
        const double a = 0x1.4f2ec413cb52ap-1;  // sqrt(3/7)
        g[0] = -1.0;
        g[1] = -a;
        g[ORDER / 2] = 0.0;
        g[ORDER - 1] = a;
        g[ORDER] = 1.0;
    }
}

// 1-D Lagrange values and derivatives: product formulas in a fixed loop order with precomputed
// reciprocals of the node differences (compile-time constants here), identical to the oracle:
//   e[i][m] = (x - g[m]) * (1 / (g[i] - g[m]));  l[i] = prod_{m != i} e[i][m];
//   dl[i] = sum_{m != i} (1/(g[i]-g[m])) * prod_{q != i,m} e[i][q]
template <int ORDER>
__device__ __forceinline__ void lagrange_1d(const double (&g)[ORDER + 1], double x, double (&l)[ORDER + 1],
                                            double (&dl)[ORDER + 1])
{
    constexpr int n = ORDER + 1;
    double inv[n][n], e[n][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int m = 0; m < n; ++m)
            if (m != i) {
                inv[i][m] = 1.0 / (g[i] - g[m]);
                e[i][m] = (x - g[m]) * inv[i][m];
            }
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double v = 1.0;
#pragma unroll
        for (int m = 0; m < n; ++m)
            if (m != i) v = v * e[i][m];
        l[i] = v;
        double d = 0.0;
#pragma unroll
        for (int m = 0; m < n; ++m) {
            if (m == i) continue;
            double t = inv[i][m];
#pragma unroll
            for (int q = 0; q < n; ++q)
                if (q != i && q != m) t = t * e[i][q];
            d = d + t;
        }
        dl[i] = d;
    }
}

// The node loads of the unrolled sum-factorised loops are kept from being hoisted to the top (all (order+1)^3 x 3
// values live at once: spills) by a compiler fence every MM_GLL_FENCE_EVERY rows of nodes (cfg5's locate stage, ms:
// every row 5.04, every 2nd 4.92, every 3rd / 5th 5.06 / 5.03, none: 470 registers spilled).
#ifndef MM_GLL_FENCE_EVERY
#define MM_GLL_FENCE_EVERY 2
#endif
#define MM_GLL_ROW_FENCE(row) do { if ((row) % MM_GLL_FENCE_EVERY == 0) asm volatile("" ::: "memory"); } while (0)
// ... and the running sums pinned there as well (round 4): a memory fence keeps the LOADS of a row behind it but not
// their consumption -- the scheduler still parks every loaded value in a register and folds them in at the end.
template <int N>
__device__ __forceinline__ void gll_pin(double (&v)[N])
{
#pragma unroll
    for (int q = 0; q < N; ++q) asm volatile("" : "+v"(v[q])::"memory");
}
// Waves per SIMD the register allocator must leave room for in the GLL locate kernels.  1: it takes what the kernel
// needs -- at order 4 in 3-D 256 VGPRs + ~90 AGPRs, one wave per SIMD, nothing in scratch memory -- and the lower orders
// still run 2 to 5 waves (95 - 211 VGPRs).  Asking for 3 (168 VGPRs, rounds 1 - 2) left 340 registers of the order-4
// kernels in scratch: cfg5's locate stage 6.2 ms at 3, 5.5 at 2 (150 spilled), 5.1 at 1.
#ifndef MM_GLL_WAVES
#define MM_GLL_WAVES 1
#endif
#ifndef MM_GLL_PIN_SUMS   // 1: the partial sums of the MAP are pinned at every row of nodes as well (see gll_pin).  Measured at
#define MM_GLL_PIN_SUMS 0 //    cfg5's shape and left off: it serialises the walking passes' node loads (0.46 -> 1.0 ms a pass)
#endif
#ifndef MM_GLL_VALUES_WAVES   // waves per SIMD gll_values_kernel is compiled for
#define MM_GLL_VALUES_WAVES 2
#endif
#ifndef MM_GLL_GUESS_TRIPS   // (tuning builds only: the oracle's start runs 8)
#define MM_GLL_GUESS_TRIPS 8
#endif
constexpr int kGllGuessTrips = MM_GLL_GUESS_TRIPS;   // hex8 trips of the corner solve that starts a 3-D inverse transform
constexpr double kGllGuessMax = 3.0;    // a start beyond this (or NaN) is not used

template <int ORDER, int DIM>
struct Gll {
    static constexpr int n = ORDER + 1;
    static constexpr int P = DIM == 3 ? n * n * n : n * n;

    // reference coordinates of pnt inside the element whose control nodes start at ctrl
    // ([P][DIM]); NaNs when the iteration fails
    static __device__ __forceinline__ void inverse_transform(const double (&pnt)[DIM],
                                                             const double *__restrict__ ctrl, double (&xi)[DIM])
    {
        double g[n];
        gll_nodes<ORDER>(g);
#pragma unroll
        for (int d = 0; d < DIM; ++d) xi[d] = 0.0;
        if (DIM == 3 && ORDER >= 2) {
            // Start from the solution of the element's eight CORNERS' trilinear map (newton_hex8_start of
            // mm_newton_hex8.h: at most kGllGuessTrips trips of ~80 fp64 instructions -- the polynomial form of the map,
            // round 4; rounds 2-3 ran the reference-order hex8 iteration here, ~300 a trip -- against ~1.2 k for one step
            // here): a straight-sided element -- its nodes the trilinear images of the GLL points -- is then left
            // after one or two steps instead of five, a curved one after three.  Same arithmetic as the oracle's start
            // (mmo_gll_inverse_transform -> mmo_hex8_start; bit-identical: tests/test_newton_host.py).  A start
            // that is not finite or far outside is not used.
            double cx[8], cy[8], cz[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int node = (MM_R(c) > 0 ? n - 1 : 0) + n * ((MM_S(c) > 0 ? n - 1 : 0) + n * (MM_T(c) > 0 ? n - 1 : 0));
                cx[c] = ctrl[3 * node + 0];
                cy[c] = ctrl[3 * node + 1];
                cz[c] = ctrl[3 * node + 2];
            }
            double q[3];
#ifdef MM_GLL_OLD_START   // timing experiment only (the oracle follows newton_hex8_start)
            (void)newton_hex8<true>(pnt[0], pnt[1], pnt[DIM - 1], cx, cy, cz, q, kGllGuessTrips);
#else
            newton_hex8_start(pnt[0], pnt[1], pnt[DIM - 1], cx, cy, cz, q, kGllGuessTrips);
#endif
            if (fabs(q[0]) <= kGllGuessMax && fabs(q[1]) <= kGllGuessMax && fabs(q[2]) <= kGllGuessMax) {
                xi[0] = q[0];
                xi[1] = q[1];
                xi[DIM - 1] = q[2];
            }
        }
        for (int it = 0; it < 25; ++it) {
            double l[DIM][n], dl[DIM][n];
#pragma unroll
            for (int d = 0; d < DIM; ++d) lagrange_1d<ORDER>(g, xi[d], l[d], dl[d]);
            double x[DIM], J[DIM][DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                x[a] = 0.0;
#pragma unroll
                for (int b = 0; b < DIM; ++b) J[a][b] = 0.0;
            }
            // Tensor-product (sum-factorised) evaluation of the map and its Jacobian, in exactly the
            // oracle's order of partial sums (mmo_gll_inverse_transform): innermost over i with l0 /
            // dl0, then over j, then over k -- and every partial sum accumulated with ONE fused multiply-add
            // (this path's arithmetic is our own definition, the oracle uses fma() at the same places):
            // ~1.2 k fp64 instructions per step at order 4 instead of 2.1 k, a third of the live registers
            // of the plain triple sum.
            if (DIM == 3) {
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    double b00[3] = {0.0, 0.0, 0.0}, b01[3] = {0.0, 0.0, 0.0}, b10[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        // keep the scheduler from hoisting all (order+1)^3 node loads to the top of
                        // the unrolled loop: one row of nodes at a time
                        MM_GLL_ROW_FENCE(j);
                        if (MM_GLL_PIN_SUMS) {
                            gll_pin(b00);
                            gll_pin(b01);
                            gll_pin(b10);
                            gll_pin(x);
                        }
                        double a0[3] = {0.0, 0.0, 0.0}, a1[3] = {0.0, 0.0, 0.0};
#pragma unroll
                        for (int i = 0; i < n; ++i) {
                            const double *X = ctrl + 3 * (i + n * (j + n * k));
#pragma unroll
                            for (int a = 0; a < 3; ++a) {
                                const double Xa = X[a];
                                a0[a] = __builtin_fma(l[0][i], Xa, a0[a]);
                                a1[a] = __builtin_fma(dl[0][i], Xa, a1[a]);
                            }
                        }
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            b00[a] = __builtin_fma(l[1][j], a0[a], b00[a]);
                            b01[a] = __builtin_fma(dl[1][j], a0[a], b01[a]);
                            b10[a] = __builtin_fma(l[1][j], a1[a], b10[a]);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        x[a] = __builtin_fma(l[DIM - 1][k], b00[a], x[a]);
                        J[a][0] = __builtin_fma(l[DIM - 1][k], b10[a], J[a][0]);
                        J[a][1] = __builtin_fma(l[DIM - 1][k], b01[a], J[a][1]);
                        J[a][DIM - 1] = __builtin_fma(dl[DIM - 1][k], b00[a], J[a][DIM - 1]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < n; ++j) {
                    double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < n; ++i) {
                        const double *X = ctrl + 2 * (i + n * j);
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const double Xa = X[a];
                            a0[a] = __builtin_fma(l[0][i], Xa, a0[a]);
                            a1[a] = __builtin_fma(dl[0][i], Xa, a1[a]);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        x[a] = __builtin_fma(l[1][j], a0[a], x[a]);
                        J[a][0] = __builtin_fma(l[1][j], a1[a], J[a][0]);
                        J[a][1] = __builtin_fma(dl[1][j], a0[a], J[a][1]);
                    }
                }
            }
            double r[DIM], dxi[DIM];
#pragma unroll
            for (int a = 0; a < DIM; ++a) r[a] = x[a] - pnt[a];
            if (DIM == 3) {
                const double c00 = J[1][1] * J[DIM - 1][DIM - 1] - J[1][DIM - 1] * J[DIM - 1][1];
                const double c01 = J[1][DIM - 1] * J[DIM - 1][0] - J[1][0] * J[DIM - 1][DIM - 1];
                const double c02 = J[1][0] * J[DIM - 1][1] - J[1][1] * J[DIM - 1][0];
                const double det = (J[0][0] * c00 + J[0][1] * c01) + J[0][DIM - 1] * c02;
                const double rdet = 1.0 / det;
                const double i00 = c00 * rdet;
                const double i01 = (J[0][DIM - 1] * J[DIM - 1][1] - J[0][1] * J[DIM - 1][DIM - 1]) * rdet;
                const double i02 = (J[0][1] * J[1][DIM - 1] - J[0][DIM - 1] * J[1][1]) * rdet;
                const double i10 = c01 * rdet;
                const double i11 = (J[0][0] * J[DIM - 1][DIM - 1] - J[0][DIM - 1] * J[DIM - 1][0]) * rdet;
                const double i12 = (J[0][DIM - 1] * J[1][0] - J[0][0] * J[1][DIM - 1]) * rdet;
                const double i20 = c02 * rdet;
                const double i21 = (J[0][1] * J[DIM - 1][0] - J[0][0] * J[DIM - 1][1]) * rdet;
                const double i22 = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rdet;
                dxi[0] = (i00 * r[0] + i01 * r[1]) + i02 * r[DIM - 1];
                dxi[1] = (i10 * r[0] + i11 * r[1]) + i12 * r[DIM - 1];
                dxi[DIM - 1] = (i20 * r[0] + i21 * r[1]) + i22 * r[DIM - 1];
            } else {
                const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
                const double rdet = 1.0 / det;
                dxi[0] = (J[1][1] * r[0] - J[0][1] * r[1]) * rdet;
                dxi[1] = (J[0][0] * r[1] - J[1][0] * r[0]) * rdet;
            }
            double step = 0.0;
            bool bad = false;
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
                xi[a] = xi[a] - dxi[a];
                if (fabs(dxi[a]) > step) step = fabs(dxi[a]);
                if (!(fabs(xi[a]) <= 10.0)) bad = true;  // also catches NaN
            }
            if (bad) break;
            if (step < 1e-10) return;   // (the update just applied: what is left is its square)
        }
#pragma unroll
        for (int d = 0; d < DIM; ++d) xi[d] = NAN;
    }

    static __device__ __forceinline__ void coefficients(const double (&xi)[DIM], double *__restrict__ out)
    {
        double g[n];
        gll_nodes<ORDER>(g);
        double l[DIM][n], dl[DIM][n];
#pragma unroll
        for (int d = 0; d < DIM; ++d) lagrange_1d<ORDER>(g, xi[d], l[d], dl[d]);
        if (DIM == 3) {
#pragma unroll
            for (int k = 0; k < n; ++k)
#pragma unroll
                for (int j = 0; j < n; ++j)
#pragma unroll
                    for (int i = 0; i < n; ++i) out[i + n * (j + n * k)] = (l[0][i] * l[1][j]) * l[DIM - 1][k];
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j)
#pragma unroll
                for (int i = 0; i < n; ++i) out[i + n * j] = l[0][i] * l[1][j];
        }
    }

    // sum_p coeff[p] * f[p] in NumPy's row-sum order (np.sum(coeffs * field[elem], axis=1), reference
    // interpolator.py:976; the order is spelled out in mm_gather.hip), coeff[p] formed exactly as
    // coefficients() forms it -- or 0.0 for a point that was not found (`zero`).  One lane, all P
    // terms: the values-only pipeline uses this at the point of acceptance instead of writing the
    // P coefficients (1 kB per target at order 4) for a gather kernel to read back.
    // LEAN (gll_values_kernel): the innermost 1-D values are made opaque at every row, so that the row's coefficient
    // products cannot be formed before it -- left alone the scheduler forms all P of them first (250 registers at
    // order 4 in 3-D: one wave per SIMD)
    template <bool LEAN = false>
    static __device__ __forceinline__ double weighted_sum(const double (&l)[DIM][n], bool zero,
                                                          const double *__restrict__ f)
    {
        constexpr int tail = P & 7, nfull = P - tail;
        double r[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
        double res = 0.;
        constexpr int nk = DIM == 3 ? n : 1;
        double m[n];
#pragma unroll
        for (int i = 0; i < n; ++i) m[i] = l[0][i];
#pragma unroll
        for (int k = 0; k < nk; ++k)
#pragma unroll
            for (int j = 0; j < n; ++j) {
                // one row of field values at a time (see inverse_transform): hoisting all P loads
                // to the top of the unrolled loop spills
                if (LEAN) {
                    // (... and the running sums are pinned at every row: without that the scheduler keeps every loaded
                    // field value in a register and adds them all up at the end)
#pragma unroll
                    for (int i = 0; i < n; ++i) asm volatile("" : "+v"(m[i])::"memory");
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(r[q])::"memory");
                    asm volatile("" : "+v"(res)::"memory");
                } else {
                    MM_GLL_ROW_FENCE(j);
                }
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    const int p = i + n * (j + n * k);
                    double c = DIM == 3 ? (m[i] * l[1][j]) * l[DIM - 1][k] : m[i] * l[1][j];
                    if (zero) c = 0.0;
                    const double a = c * f[p];
                    if (P < 8) {
                        res += a;
                    } else if (p < 8) {
                        r[p] = a;
                    } else if (p < nfull) {
                        r[p & 7] += a;
                    }
                    if (P >= 8 && p == nfull - 1)
                        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                    if (P >= 8 && p >= nfull) res += a;
                }
            }
        return 0.0 + res;   // NumPy starts the reduction from the identity (-0.0 reads +0.0)
    }
};

// What a located target leaves behind: element id and coefficients (the staged mm_locate_gll), and /
// or the interpolated values themselves (mm_interpolate_gll without operator outputs).
struct GllEmit {
    i64 *elem;             // [N] or null
    double *coeffs;        // [N][P] or null
    const double *fields;  // [ncomp][nelem][P] or null
    double *out;           // [N][ncomp] or null
    int ncomp;
    // Deferred values (round 4): with xi_defer the locate kernels only leave {element or -1, reference coordinates}
    // per target and gll_values_kernel forms the weighted sums afterwards -- the same arithmetic in a kernel of its
    // own.  Inside the order-4 locate kernels the unrolled 125-term sum cost 1.4 of the stage's 4.4 ms at cfg5's
    // shape and, worse, registers: those kernels run ONE wave per SIMD (256 VGPRs + ~120 AGPRs).
    double *xi_defer = nullptr;   // [N][DIM] or null
    int *elem_defer = nullptr;    // [N]
};

// found == false: the reference's "-1 and zero coefficients"; NumPy's field[-1] is the LAST element
template <int ORDER, int DIM, bool DEFER = false>
__device__ __forceinline__ void gll_emit(const GllEmit &em, i64 i, i64 e, const double (&xi)[DIM], bool found, i64 nelem)
{
    if (DEFER) {   // (compile-time: the instance carries neither the coefficient products nor the sums)
        em.elem_defer[i] = found ? (int)e : -1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) em.xi_defer[i * DIM + d] = found ? xi[d] : 0.0;
        return;
    }
    using G = Gll<ORDER, DIM>;
    constexpr int P = G::P;
    if (em.elem) em.elem[i] = found ? e : -1;
    if (em.coeffs) {
        if (found) G::coefficients(xi, em.coeffs + i * P);
        else
            for (int p = 0; p < P; ++p) em.coeffs[i * P + p] = 0.0;
    }
    if (em.out) {
        double g[G::n];
        gll_nodes<ORDER>(g);
        double l[DIM][G::n], dl[DIM][G::n];
#pragma unroll
        for (int d = 0; d < DIM; ++d) lagrange_1d<ORDER>(g, found ? xi[d] : 0.0, l[d], dl[d]);
        const i64 ef = found ? e : nelem - 1;
        for (int c = 0; c < em.ncomp; ++c)
            em.out[i * em.ncomp + c] = G::weighted_sum(l, !found, em.fields + ((i64)c * nelem + ef) * P);
    }
}

// The deferred half of gll_emit (GllEmit::xi_defer): out[i][c] = sum_p coeff_p(xi_i) * field[c][elem_i][p], the
// coefficients and the sum formed exactly as there (NumPy's row-sum order; a target that was not found reads the LAST
// element with zero coefficients like NumPy's field[-1] * 0).  One lane per target; neighbouring targets lie in the same
// or neighbouring elements, so a wave's loads of a field row fall on a few lines.
template <int ORDER, int DIM>
__global__ __launch_bounds__(256, MM_GLL_VALUES_WAVES) void gll_values_kernel(i64 npoints, const int *__restrict__ elem,
                                                         const double *__restrict__ xi_all,
                                                         const double *__restrict__ fields, i64 nelem, int ncomp,
                                                         double *__restrict__ out)
{
    using G = Gll<ORDER, DIM>;
    constexpr int P = G::P;
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoints) return;
    const int e = elem[i];
    const bool found = e >= 0;
    double g[G::n];
    gll_nodes<ORDER>(g);
    double l[DIM][G::n], dl[DIM][G::n];
#pragma unroll
    for (int d = 0; d < DIM; ++d) lagrange_1d<ORDER>(g, found ? xi_all[i * DIM + d] : 0.0, l[d], dl[d]);
    const i64 ef = found ? (i64)e : nelem - 1;
    for (int c = 0; c < ncomp; ++c) {
        // (the 125 coefficient products do not depend on c: left alone, the compiler forms them all in front of this loop
        // and keeps them in 250 registers -- one wave per SIMD; weighted_sum<LEAN> forms them row by row)
        out[i * ncomp + c] = G::template weighted_sum<true>(l, !found, fields + ((i64)c * nelem + ef) * P);
    }
}

// Control flow of reference interpolator.py:1181-1233 (see the oracle's mmo_locate_gll), scheduled
// as COMPACTING PASSES like the hex8 locate: a pass performs at most one inverse transform per
// still-open target (candidate j of its list) and re-queues the unresolved ones densely as (target,
// j+1), so a wave never waits for its unluckiest lane's whole candidate walk.  k passes are launched
// (each advances every open target by at least one candidate), the late ones over a nearly empty
// queue.  With snap_to_nearest the least-outside candidate seen so far travels in per-target state
// arrays.  Re-queue entries are batched per wave in LDS (one global atomic per ~200 entries).
constexpr int kGllWaveQueue = 256;
constexpr int kGllLazyK = 8;   // candidates asked of the kNN stage up front by mm_interpolate_gll
constexpr int kGllWalkFrom = 1; // passes that advance one candidate before the lanes walk their lists (round 4: 1 -- with the
                                // values formed in a kernel of their own the one-candidate passes 1 and 2 cost more than they save: 3.12 -> 2.92 ms)

template <int ORDER, int DIM, typename IDX, bool DEFER = false>
__global__ __launch_bounds__(64, MM_GLL_WAVES) void locate_gll_pass_kernel(i64 k, int kavail, i64 npoints,
                                                             const IDX *__restrict__ nn,
                                                             const double *__restrict__ gll_points, i64 nelem,
                                                             const double *__restrict__ points, double tolerance,
                                                             int snap_to_nearest, GllEmit em,
                                                             unsigned long long *__restrict__ nmissing,
                                                             const int *__restrict__ order,
                                                             const int2 *__restrict__ q_in,
                                                             const int *__restrict__ q_in_count,
                                                             int2 *__restrict__ q_out, int *__restrict__ q_out_count,
                                                             double *__restrict__ best_state,   // [N][DIM+1]
                                                             i64 *__restrict__ best_elem_state,  // [N]
                                                             int walk)
{
    using G = Gll<ORDER, DIM>;
    constexpr int P = G::P;
    __shared__ int2 s_queue[kGllWaveQueue];
    const int lane = threadIdx.x;
    int held = 0;
    unsigned long long missing_total = 0;

    const i64 total = q_in ? (i64)*q_in_count : npoints;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 first = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 trips = (total + stride - 1) / stride;
    for (i64 trip = 0; trip < trips; ++trip) {
        const i64 q = first + trip * stride;
        const bool active = q < total;
        bool requeue = false, missing = false;
        i64 i = 0;
        int j = 0;
        if (active) {
            if (q_in) {
                const int2 e = q_in[q];
                i = e.x;
                j = e.y;
            } else {
                i = order ? (i64)order[q] : q;
            }
            double pnt[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d) pnt[d] = points[i * DIM + d];
            // least-outside candidate so far (fresh in the first pass)
            double best_xi[DIM];
            double best_val = 10e9;
            i64 best_elem = 0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) best_xi[d] = 10e9;
            if (snap_to_nearest && q_in) {
                best_val = best_state[i * (DIM + 1)];
#pragma unroll
                for (int d = 0; d < DIM; ++d) best_xi[d] = best_state[i * (DIM + 1) + 1 + d];
                best_elem = best_elem_state[i];
            }
            bool found = false;
            do {   // one candidate, or (walk) one after the other until the point is found or the list ends
                // next valid candidate; this array holds the first kavail <= k of the target's list
                while (j < kavail) {
                    const i64 e = (i64)nn[i * kavail + j];
                    if (e >= 0 && e < nelem) break;
                    ++j;
                }
                if (j >= kavail) break;
                const i64 e = (i64)nn[i * kavail + j];
                double xi[DIM];
                G::inverse_transform(pnt, gll_points + e * (i64)(P * DIM), xi);
                bool isnan_any = false;
                double worst = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    if (xi[d] != xi[d]) isnan_any = true;
                    if (fabs(xi[d]) > worst) worst = fabs(xi[d]);
                }
                if (!isnan_any) {
                    if (worst < best_val) {
                        best_val = worst;
                        best_elem = e;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) best_xi[d] = xi[d];
                    }
                    bool inside = true;
#pragma unroll
                    for (int d = 0; d < DIM; ++d)
                        if (!(fabs(xi[d]) < tolerance)) inside = false;
                    if (inside) {
                        gll_emit<ORDER, DIM, DEFER>(em, i, e, xi, true, nelem);
                        found = true;
                    }
                }
                ++j;
            } while (walk && !found);
            if (!found) {
                if (j < k) {
                    requeue = true;   // also: the short list is used up and the full one is needed
                    if (snap_to_nearest) {
                        best_state[i * (DIM + 1)] = best_val;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) best_state[i * (DIM + 1) + 1 + d] = best_xi[d];
                        best_elem_state[i] = best_elem;
                    }
                } else if (snap_to_nearest) {
#pragma unroll
                    for (int d = 0; d < DIM; ++d) {
                        double v = best_xi[d];
                        if (v < -1.02) v = -1.02;
                        if (v > 1.02) v = 1.02;
                        best_xi[d] = v;
                    }
                    gll_emit<ORDER, DIM, DEFER>(em, i, best_elem, best_xi, true, nelem);
                } else {
                    gll_emit<ORDER, DIM, DEFER>(em, i, 0, best_xi, false, nelem);
                    missing = true;
                }
            }
        }
        missing_total += __popcll(__ballot(missing));
        const unsigned long long vote = __ballot(requeue);
        if (requeue) s_queue[held + __popcll(vote & ((1ull << lane) - 1ull))] = make_int2((int)i, j);
        held += __popcll(vote);
        if (held > kGllWaveQueue - 64 || (trip == trips - 1 && held > 0)) {
            int base = 0;
            if (lane == 0) base = atomicAdd(q_out_count, held);
            base = __shfl(base, 0);
            for (int t = lane; t < held; t += 64) q_out[base + t] = s_queue[t];
            held = 0;
        }
    }
    if (lane == 0 && missing_total) atomicAdd(nmissing, missing_total);
}

// First pass with the control nodes in LDS.  Targets arrive sorted by their first candidate element
// (~90 per element at cfg5's shape), so the 64 lanes of a wave need one to three elements: up to kStaged
// of them are copied into LDS once per solve (coalesced) and every Newton step reads them from there -- all
// lanes of an element the same address (broadcast) -- instead of going back to L1/L2 for 3 KB per step.  A
// wave with more distinct elements (thinly populated elements) takes further turns of the stage/solve
// loop.  Same arithmetic, same results as locate_gll_pass_kernel with q_in == null.
template <int ORDER, int DIM, typename IDX, bool DEFER = false>
__global__ __launch_bounds__(64, MM_GLL_WAVES) void locate_gll_first_pass_kernel(
    i64 k, int kavail, i64 npoints, const IDX *__restrict__ nn, const double *__restrict__ gll_points, i64 nelem,
    const double *__restrict__ points, double tolerance, int snap_to_nearest, GllEmit em,
    unsigned long long *__restrict__ nmissing, const int *__restrict__ order,
    int2 *__restrict__ q_out, int *__restrict__ q_out_count, double *__restrict__ best_state,
    i64 *__restrict__ best_elem_state)
{
    // (Round 4 also sent the LATER passes through this kernel, {target, next candidate} queue entries in the visiting
    // order: slower -- a wave of open targets spans many more than two elements and takes a turn of the stage / solve
    // loop for every pair: 0.9 ms a pass at cfg5's shape against 0.4 for the per-lane kernel.  Taken out again.)
    using G = Gll<ORDER, DIM>;
    constexpr int P = G::P;
    constexpr int kNodeDoubles = P * DIM;
    __shared__ int2 s_queue[kGllWaveQueue];
    // elements staged per turn: two at order 4 in 3-D (3 KB each; four measured slower: 7.9 vs 7.6 ms at cfg5's
    // shape -- a tenth of the waves span more than two elements, every wave pays the LDS), more for small elements
    constexpr int kStaged = kNodeDoubles <= 81 ? 4 : 2;
    __shared__ double s_ctrl[kStaged][kNodeDoubles];
    const int lane = threadIdx.x;
    int held = 0;
    unsigned long long missing_total = 0;

    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 first = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 trips = (npoints + stride - 1) / stride;
    for (i64 trip = 0; trip < trips; ++trip) {
        const i64 q = first + trip * stride;
        const bool active = q < npoints;
        bool requeue = false, missing = false;
        i64 i = 0, e = -1;
        int j = 0;
        double pnt[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) pnt[d] = 0.0;
        if (active) {
            i = order ? (i64)order[q] : q;
#pragma unroll
            for (int d = 0; d < DIM; ++d) pnt[d] = points[i * DIM + d];
            // first valid candidate
            while (j < kavail) {
                e = (i64)nn[i * kavail + j];
                if (e >= 0 && e < nelem) break;
                ++j;
            }
        }
        bool pending = active && j < kavail;
        bool found = false;
        double best_xi[DIM];
        double best_val = 10e9;
        i64 best_elem = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) best_xi[d] = 10e9;
        while (__any(pending)) {
            // the (up to) kStaged elements of this turn: the first pending lane's, the first other one, ...
            int which = -1;   // this lane's element among the staged ones (an offset into s_ctrl)
            unsigned long long left = __ballot(pending);
#pragma unroll
            for (int m = 0; m < kStaged; ++m) {
                if (left == 0) break;   // (wave-uniform)
                const i64 em = __shfl(e, __ffsll((long long)left) - 1);
                for (int t = lane; t < kNodeDoubles; t += 64) s_ctrl[m][t] = gll_points[em * (i64)kNodeDoubles + t];
                const bool hit = pending && e == em;
                if (hit) which = m * kNodeDoubles;
                left &= ~__ballot(hit);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool mine = which >= 0;
            if (mine) {
                double xi[DIM];
                // a per-lane OFFSET into the staged nodes: with a select between the arrays the compiler
                // reads every node of all of them and picks the values lane by lane (750 v_cndmask_b32 and
                // twice the LDS reads per Newton step at order 4 with two arrays)
                asm volatile("" : "+v"(which));
                G::inverse_transform(pnt, &s_ctrl[0][0] + which, xi);
                bool isnan_any = false;
                double worst = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    if (xi[d] != xi[d]) isnan_any = true;
                    if (fabs(xi[d]) > worst) worst = fabs(xi[d]);
                }
                if (!isnan_any) {
                    if (worst < best_val) {
                        best_val = worst;
                        best_elem = e;
#pragma unroll
                        for (int d = 0; d < DIM; ++d) best_xi[d] = xi[d];
                    }
                    bool inside = true;
#pragma unroll
                    for (int d = 0; d < DIM; ++d)
                        if (!(fabs(xi[d]) < tolerance)) inside = false;
                    if (inside) {
                        gll_emit<ORDER, DIM, DEFER>(em, i, e, xi, true, nelem);
                        found = true;
                    }
                }
                ++j;
                pending = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // every lane is done with the staged nodes
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (active && !found) {
            if (j < k) {
                requeue = true;
                if (snap_to_nearest) {
                    best_state[i * (DIM + 1)] = best_val;
#pragma unroll
                    for (int d = 0; d < DIM; ++d) best_state[i * (DIM + 1) + 1 + d] = best_xi[d];
                    best_elem_state[i] = best_elem;
                }
            } else if (snap_to_nearest) {
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    double v = best_xi[d];
                    if (v < -1.02) v = -1.02;
                    if (v > 1.02) v = 1.02;
                    best_xi[d] = v;
                }
                gll_emit<ORDER, DIM, DEFER>(em, i, best_elem, best_xi, true, nelem);
            } else {
                gll_emit<ORDER, DIM, DEFER>(em, i, 0, best_xi, false, nelem);
                missing = true;
            }
        }
        missing_total += __popcll(__ballot(missing));
        const unsigned long long vote = __ballot(requeue);
        if (requeue) s_queue[held + __popcll(vote & ((1ull << lane) - 1ull))] = make_int2((int)i, j);
        held += __popcll(vote);
        if (held > kGllWaveQueue - 64 || (trip == trips - 1 && held > 0)) {
            int base = 0;
            if (lane == 0) base = atomicAdd(q_out_count, held);
            base = __shfl(base, 0);
            for (int t = lane; t < held; t += 64) q_out[base + t] = s_queue[t];
            held = 0;
        }
    }
    if (lane == 0 && missing_total) atomicAdd(nmissing, missing_total);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Element-nodal gather with NumPy's row-sum order (see mm_gather.hip): 8 lanes per target, lane j
// owns the running partial r[j]; the node "ids" are implicit: elem * P + p.
template <bool POINT_MAJOR>
__global__ __launch_bounds__(256) void gather_elem_kernel(const double *__restrict__ fields, i64 nelem, int ncomp,
                                                          const i64 *__restrict__ elem,
                                                          const double *__restrict__ coeffs, i64 npoints, int P,
                                                          double *__restrict__ out)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n_raw = t >> 3;
    const int j = (int)(t & 7);
    const int group_base = (threadIdx.x & 63) & ~7;
    const bool valid = n_raw < npoints;
    const i64 n = valid ? n_raw : npoints - 1;
    const i64 e_raw = elem[n];
    // -1 (not found, zero coefficients) reads the LAST element like NumPy's field[-1]: the sign of the zero
    const i64 e = (unsigned long long)e_raw < (unsigned long long)nelem ? e_raw
                  : (e_raw < 0 && e_raw >= -nelem ? e_raw + nelem : 0);
    const double *crow = coeffs + n * P;
    const int tail = P & 7;
    const int nfull = P - tail;
    for (int c = 0; c < ncomp; ++c) {
        const double *f = fields + ((i64)c * nelem + e) * P;
        double res;
        if (P < 8) {
            const double a = j < P ? crow[j] * f[j] : 0.0;
            res = 0.;
            for (int i = 0; i < P; ++i) res += __shfl(a, group_base + i);
        } else {
            double r = crow[j] * f[j];
            for (int i = 8; i < nfull; i += 8) r += crow[i + j] * f[i + j];
            double s = r + dpp_f64<0xB1>(r);
            s = s + dpp_f64<0x4E>(s);
            s = s + dpp_f64<0x141>(s);
            const double a = j < tail ? crow[nfull + j] * f[nfull + j] : 0.0;
            res = s;
            for (int i = 0; i < tail; ++i) res += __shfl(a, group_base + i);
        }
        if (valid && j == 0) {
            // NumPy starts a reduction from the identity: 0.0 + (row sum), visible where the sum is -0.0
            if (POINT_MAJOR) out[n * ncomp + c] = 0.0 + res;
            else out[(i64)c * npoints + n] = 0.0 + res;
        }
    }
}

// Lazily evaluated candidate lists (fused pipeline): nn holds only the kavail nearest; after kavail
// passes the still-open targets get their full lists from the generic kNN kernel (list mode) and
// the remaining passes read those.  The k' nearest are the first k' of the k nearest, so results
// do not depend on it.
struct GllLazy {
    const mm_knn_index *index;
    int *nn_full;   // [N][k], rows filled on demand
};

__global__ __launch_bounds__(256) void gll_queue_ids_kernel(const int2 *__restrict__ q, const int *__restrict__ q_count,
                                                            int *__restrict__ list)
{
    const i64 total = *q_count;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) list[t] = q[t].x;
}

template <int ORDER, int DIM, typename IDX, bool DEFER = false>
int launch_locate(mm_context *ctx, i64 k, int kavail, i64 npoints, const IDX *nn, const double *gll, i64 nelem,
                  const double *pts, double tol, int snap, const GllEmit &em, unsigned long long *nmiss,
                  const int *order, int2 *qa, int2 *qb, int *counters, double *best_state, i64 *best_elem_state,
                  const GllLazy *lazy, int *id_list)
{
    const i64 full_grid = (npoints + 63) / 64;
    // Pass c reads queue c (pass 0: every target) and advances each open target by ONE candidate -- so that a
    // wave never waits for its unluckiest lane while most lanes still have work -- until pass kGllWalkFrom: by
    // then the queue is a sliver of the targets (mostly points that lie outside every candidate and will use
    // up their list) and each further launch would cost a full Newton solve of latency over a nearly empty
    // chip (17 such passes: 2.9 of cfg5's 8.6 ms), so from there a lane walks its list to the end.  With lazy
    // lists the walk ends at the short list's end, the open targets' full lists are fetched once, and one
    // more walking pass finishes them.
    static const int walk_from = getenv("MM_GLL_WALK_FROM") ? atoi(getenv("MM_GLL_WALK_FROM")) : kGllWalkFrom;
    i64 jdone = 0;        // candidates every open target is past
    bool full = false;    // the passes read the full lists (lazy only)
    for (int c = 0; c == 0 || jdone < k; ++c) {
        const int2 *q_in = c == 0 ? nullptr : ((c & 1) ? qa : qb);
        int2 *q_out = (c & 1) ? qb : qa;
        // persistent waves; later passes only know their size on the device
        i64 grid = full_grid >> (c < 6 ? c : 6);
        if (grid > 16384) grid = 16384;
        if (grid < 256) grid = full_grid < 256 ? full_grid : 256;
        if (lazy && !full && jdone >= kavail) {
            hipLaunchKernelGGL(gll_queue_ids_kernel, dim3(64), dim3(256), 0, ctx->stream, q_in, counters + c, id_list);
            int rc = mm_knn_query_list_impl(ctx, lazy->index, pts, npoints, k, lazy->nn_full, id_list, counters + c, -1);
            if (rc != MM_OK) return rc;
            full = true;
        }
        const i64 avail = full ? k : (i64)kavail;
        const int walk = c >= walk_from ? 1 : 0;
        if (c == 0 && k > 0 && nelem > 0) {
            hipLaunchKernelGGL((locate_gll_first_pass_kernel<ORDER, DIM, IDX, DEFER>), dim3((unsigned)grid), dim3(64), 0,
                               ctx->stream, k, kavail, npoints, nn, gll, nelem, pts, tol, snap, em, nmiss, order, q_out,
                               counters + c + 1, best_state, best_elem_state);
        } else if (full) {
            hipLaunchKernelGGL((locate_gll_pass_kernel<ORDER, DIM, int, DEFER>), dim3((unsigned)grid), dim3(64), 0, ctx->stream,
                               k, (int)k, npoints, (const int *)lazy->nn_full, gll, nelem, pts, tol, snap, em, nmiss,
                               (const int *)nullptr, q_in, counters + c, q_out, counters + c + 1, best_state,
                               best_elem_state, walk);
        } else {
            hipLaunchKernelGGL((locate_gll_pass_kernel<ORDER, DIM, IDX, DEFER>), dim3((unsigned)grid), dim3(64), 0, ctx->stream,
                               k, kavail, npoints, nn, gll, nelem, pts, tol, snap, em, nmiss, c == 0 ? order : nullptr,
                               q_in, c == 0 ? nullptr : counters + c, q_out, counters + c + 1, best_state,
                               best_elem_state, walk);
        }
        jdone = (walk && c > 0) || (walk && !(k > 0 && nelem > 0)) ? avail : jdone + 1;
    }
    return MM_OK;
}

// ---- variant 1 (reference interpolator.py:1350-1367, :1409-1473; oracle mmo_locate_gll_v1) ----------
// Per element: axis-aligned bounding box of the control nodes and their mean (sequential sum from 0.0
// in node order, divided by P) -- boxes[e] = {min[dim], max[dim], centre[dim]}.  The reference
// recomputes them per candidate and point; they only depend on the element.
template <int ORDER, int DIM>
__global__ __launch_bounds__(256) void gll_box_kernel(i64 nelem, const double *__restrict__ gll_points,
                                                      double *__restrict__ boxes)
{
    constexpr int P = Gll<ORDER, DIM>::P;
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nelem) return;
    const double *X = gll_points + e * (i64)(P * DIM);
    double *b = boxes + e * (i64)(3 * DIM);
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        double mn = X[d], mx = X[d], sum = 0.0;
        for (int p = 0; p < P; ++p) {
            const double v = X[p * DIM + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
            sum = sum + v;
        }
        b[d] = mn;
        b[DIM + d] = mx;
        b[2 * DIM + d] = sum / (double)P;
    }
}

// One lane per target, candidates in order (lock-step: this variant is the completeness path, the
// tolerance/snap variant above is the tuned one).
template <int ORDER, int DIM>
__global__ __launch_bounds__(64, MM_GLL_WAVES) void locate_gll_v1_kernel(i64 k, i64 npoints, const i64 *__restrict__ nn,
                                                              const double *__restrict__ gll_points, i64 nelem,
                                                              const double *__restrict__ boxes,
                                                              const double *__restrict__ points,
                                                              i64 *__restrict__ elem, double *__restrict__ coeffs,
                                                              unsigned long long *__restrict__ nhard,
                                                              const int *__restrict__ order)
{
    using G = Gll<ORDER, DIM>;
    constexpr int P = G::P;
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    bool hard = false;
    if (q < npoints) {
        const i64 i = order ? (i64)order[q] : q;
        double pnt[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) pnt[d] = points[i * DIM + d];
        i64 first_inside = -1, nearest = -1;
        double nearest_d2 = INFINITY;
        bool found = false;
        double xi[DIM];
        for (i64 j = 0; j < k && !found; ++j) {
            const i64 e = nn[i * k + j];
            if (e < 0 || e >= nelem) continue;
            const double *b = boxes + e * (i64)(3 * DIM);
            bool inside = true;
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                if (!(pnt[d] >= b[d] && pnt[d] <= b[DIM + d])) inside = false;
            if (inside) {
                if (first_inside < 0) first_inside = j;
                G::inverse_transform(pnt, gll_points + e * (i64)(P * DIM), xi);
                bool ok = true;
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    if (!(fabs(xi[d]) <= 1.04)) ok = false;   // NaN fails the comparison too
                if (ok) {
                    elem[i] = e;
                    G::coefficients(xi, coeffs + i * P);
                    found = true;
                }
            } else {
                double d2 = 0.0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    const double t = pnt[d] - b[2 * DIM + d];
                    d2 = d2 + t * t;
                }
                if (d2 < nearest_d2) {
                    nearest_d2 = d2;
                    nearest = j;
                }
            }
        }
        if (!found) {
            const i64 ind = first_inside >= 0 ? first_inside : nearest;
            if (ind < 0) {
                elem[i] = -1;
                for (int p = 0; p < P; ++p) coeffs[i * P + p] = 0.0;
                hard = true;
            } else {
                const i64 e = nn[i * k + ind];
                G::inverse_transform(pnt, gll_points + e * (i64)(P * DIM), xi);
                bool isnan_any = false, far = false;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    if (xi[d] != xi[d]) isnan_any = true;
                    if (fabs(xi[d]) >= 1.04) far = true;
                }
                hard = isnan_any;
                if (isnan_any || far) {
                    const double hard_xi[3] = {0.645, -0.5, 0.22};   // reference :1468-1471
#pragma unroll
                    for (int d = 0; d < DIM; ++d) xi[d] = hard_xi[d];
                }
                elem[i] = e;
                G::coefficients(xi, coeffs + i * P);
            }
        }
    }
    const unsigned long long mask = __ballot(hard);
    if (threadIdx.x == 0 && mask) atomicAdd(nhard, (unsigned long long)__popcll(mask));
}

// visiting order: counting sort of the targets by their first candidate element
template <typename IDX>
__global__ __launch_bounds__(256) void gll_key_kernel(i64 k, i64 npoints, const IDX *__restrict__ nn, i64 nelem,
                                                      int2 *__restrict__ key_rank, int *__restrict__ counts)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < npoints;
    int key = -1;
    if (live) {
        const i64 e = (i64)nn[t * k];
        key = (unsigned long long)e < (unsigned long long)nelem ? (int)e : (int)nelem;  // invalid -> last bin
    }
    // targets in mesh order arrive in runs with the same first candidate, and same-address atomics
    // serialise in L2: one atomic per run inside the wave (as in the kNN stage's cell_count_kernel)
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const bool head = lane == 0 || key != prev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));
    const int head_lane = 63 - __clzll((long long)upto);
    const unsigned long long after = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
    int base = 0;
    if (head && live) {
        const int next_head = after ? __ffsll((long long)after) - 1 : 64;
        base = atomicAdd(&counts[key], next_head - lane);
    }
    base = __shfl(base, head_lane);
    if (live) key_rank[t] = make_int2(key, base + (lane - head_lane));
}

__global__ __launch_bounds__(256) void gll_order_kernel(i64 npoints, const int2 *__restrict__ key_rank,
                                                        const int *__restrict__ start, int *__restrict__ order)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npoints) return;
    const int2 kr = key_rank[t];
    order[start[kr.x] + kr.y] = (int)t;
}

}  // namespace

int mm_exclusive_scan_int(mm_context *ctx, const int *counts, i64 n, int *start, int *tile_sums);
int mm_knn_build_impl(mm_context *ctx, const double *src_d, i64 nsrc, i64 ndim, mm_knn_index **out,
                      bool use_context_buffers, const double *box_partial_d, int box_nblocks, bool hex8_centroids = false);
int mm_knn_query_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k, void *idx_d,
                      double *dist_d, bool idx_is_int32);

// The locate stage on ctx->stream: visiting order, passes, no synchronisation.  nn holds the first
// kavail (<= k) candidates of every target (kavail < k only with `lazy`).  The number of targets
// that were not found is added to ctx->d_counters[0].
template <typename IDX>
static int locate_gll_run(mm_context *ctx, int order, int dim, i64 k, int kavail, i64 npoints, const IDX *nn,
                          const double *gll_points_d, i64 nelem, const double *points_d, double tolerance,
                          int snap_to_nearest, const GllEmit &em, const GllLazy *lazy)
{
    unsigned long long *nm = (unsigned long long *)ctx->d_counters;
    // scratch: visiting order, two pass queues, their counters, snap state
    const int *visit = nullptr;
    const i64 nbins = nelem + 1;
    const i64 ntiles = (nbins + 1023) / 1024;
    int rc = mm_scratch_begin(ctx, 3 * mm_round256((size_t)npoints * sizeof(int2)) +
                                       2 * mm_round256((size_t)npoints * sizeof(int)) +
                                       2 * mm_round256((size_t)(nbins + 1) * sizeof(int)) +
                                       mm_round256((size_t)ntiles * sizeof(int)) +
                                       (snap_to_nearest ? mm_round256((size_t)npoints * 5 * sizeof(double)) : 0) +
                                       mm_round256(sizeof(int) * (MM_KNN_MAX_K + 8)) + 4096);
    if (rc != MM_OK) return rc;
    int2 *qa = (int2 *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int2));
    int2 *qb = (int2 *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int2));
    int *counters = (int *)mm_scratch_take(ctx, sizeof(int) * (MM_KNN_MAX_K + 8));
    int *id_list = (int *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int));
    double *best_state = snap_to_nearest ? (double *)mm_scratch_take(ctx, (size_t)npoints * 4 * sizeof(double)) : nullptr;
    i64 *best_elem_state = snap_to_nearest ? (i64 *)mm_scratch_take(ctx, (size_t)npoints * sizeof(i64)) : nullptr;
    MM_REQUIRE(qa && qb && counters && id_list && (!snap_to_nearest || (best_state && best_elem_state)),
               "scratch carve failed");
    MM_HIP_CHECK(hipMemsetAsync(counters, 0, sizeof(int) * (MM_KNN_MAX_K + 8), ctx->stream));
    if (k > 0 && nelem > 0) {
        int2 *key_rank = (int2 *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int2));
        int *ord = (int *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int));
        int *counts = (int *)mm_scratch_take(ctx, (size_t)(nbins + 1) * sizeof(int));
        int *start = (int *)mm_scratch_take(ctx, (size_t)(nbins + 1) * sizeof(int));
        int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
        MM_REQUIRE(key_rank && ord && counts && start && tile_sums, "scratch carve failed");
        MM_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)(nbins + 1) * sizeof(int), ctx->stream));
        const unsigned gp = (unsigned)((npoints + 255) / 256);
        hipLaunchKernelGGL((gll_key_kernel<IDX>), dim3(gp), dim3(256), 0, ctx->stream, (i64)kavail, npoints, nn, nelem,
                           key_rank, counts);
        rc = mm_exclusive_scan_int(ctx, counts, nbins, start, tile_sums);
        if (rc != MM_OK) return rc;
        hipLaunchKernelGGL(gll_order_kernel, dim3(gp), dim3(256), 0, ctx->stream, npoints, key_rank, start, ord);
        visit = ord;
    }
    rc = MM_OK;
    // (deferred values: only the fused pipeline's int32 lists ask for them)
    constexpr bool kCanDefer = sizeof(IDX) == sizeof(int);
    const bool defer = kCanDefer && em.xi_defer != nullptr;
#define MM_GLL_CASE(O, D)                                                                                          \
    if (order == O && dim == D) {                                                                                  \
        if (defer)                                                                                                 \
            rc = launch_locate<O, D, IDX, kCanDefer>(ctx, k, kavail, npoints, nn, gll_points_d, nelem, points_d,   \
                                                     tolerance, snap_to_nearest, em, nm, visit, qa, qb, counters,  \
                                                     best_state, best_elem_state, lazy, id_list);                  \
        else                                                                                                       \
            rc = launch_locate<O, D, IDX>(ctx, k, kavail, npoints, nn, gll_points_d, nelem, points_d, tolerance,   \
                                          snap_to_nearest, em, nm, visit, qa, qb, counters, best_state,            \
                                          best_elem_state, lazy, id_list);                                         \
    }
    MM_GLL_CASE(1, 2) MM_GLL_CASE(1, 3) MM_GLL_CASE(2, 2) MM_GLL_CASE(2, 3) MM_GLL_CASE(4, 2) MM_GLL_CASE(4, 3)
#undef MM_GLL_CASE
    if (rc != MM_OK) return rc;
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

extern "C" int64_t mm_locate_gll(mm_context *ctx, int order, int dim, int64_t k, int64_t npoints,
                                 const int64_t *nn_d, const double *gll_points_d, int64_t nelem,
                                 const double *points_d, double tolerance, int snap_to_nearest, int64_t *elem_d,
                                 double *coeffs_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(order == 1 || order == 2 || order == 4, "order must be 1, 2 or 4");
    MM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    MM_REQUIRE(k >= 0 && npoints >= 0 && nelem >= 0, "negative size");
    MM_REQUIRE(k <= MM_KNN_MAX_K, "nelem_to_search must be <= MM_KNN_MAX_K");
    MM_REQUIRE(npoints == 0 || (elem_d && coeffs_d && points_d), "null array");
    MM_REQUIRE(npoints == 0 || k == 0 || (nn_d && gll_points_d), "null array");
    MM_REQUIRE(npoints < (int64_t)0x7fffffff && nelem < (int64_t)0x7ffffff0, "too many targets / elements");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    MM_HIP_CHECK(hipMemsetAsync(ctx->d_counters, 0, sizeof(i64), ctx->stream));
    if (npoints > 0) {
        mm_stage_begin(ctx, MM_STAGE_LOCATE);
        GllEmit em = {(i64 *)elem_d, coeffs_d, nullptr, nullptr, 0};
        int rc = locate_gll_run<i64>(ctx, order, dim, k, (int)k, npoints, (const i64 *)nn_d, gll_points_d, nelem,
                                     points_d, tolerance, snap_to_nearest, em, nullptr);
        mm_stage_end(ctx, MM_STAGE_LOCATE);
        if (rc != MM_OK) return rc;
    }
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ctx->h_counters[0];
}

// Mean of the control nodes of every element, summed in node order like NumPy's mean(axis=1)
// (reference salvus_mesh_reader.py:99-100) -- the source points of the GLL path's centroid tree.
template <int DIM>
__global__ __launch_bounds__(256) void centroid_nodal_kernel(i64 nelem, int P, const double *__restrict__ gll_points,
                                                             double *__restrict__ out)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nelem) return;
    const double *row = gll_points + e * (i64)P * DIM;
    double acc[DIM];
#pragma unroll
    for (int a = 0; a < DIM; ++a) acc[a] = row[a];
    for (int p = 1; p < P; ++p)
#pragma unroll
        for (int a = 0; a < DIM; ++a) acc[a] = acc[a] + row[p * DIM + a];
#pragma unroll
    for (int a = 0; a < DIM; ++a) out[e * DIM + a] = acc[a] / (double)P;
}

// The GLL form of the whole path on resident arrays (reference interpolator.py:931-977, and the core of
// gll_2_gll :700-830): element centroids -> k nearest -> element + reference coordinates by the
// acceptance loop of :1181-1233 -> sum of coefficient x element-nodal field value.  Without operator
// outputs the weighted sum is formed where a target is accepted (no [N][P] coefficient array), and
// the candidate lists are evaluated lazily (mm_set_lazy_lists) exactly as in mm_interpolate_hex8.
extern "C" int64_t mm_interpolate_gll(mm_context *ctx, int order, int dim, const double *gll_points_d, int64_t nelem,
                                      const double *points_d, int64_t npoints, const double *fields_d,
                                      int64_t ncomp, int64_t k, double tolerance, int snap_to_nearest,
                                      double *out_d, int64_t *elem_out_d, double *coeffs_out_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(order == 1 || order == 2 || order == 4, "order must be 1, 2 or 4");
    MM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    MM_REQUIRE(k >= 1 && k <= MM_KNN_MAX_K, "nelem_to_search must be in 1..MM_KNN_MAX_K");
    MM_REQUIRE(npoints >= 0 && nelem >= 1 && ncomp >= 0, "bad sizes");
    MM_REQUIRE(npoints < (int64_t)0x7fffffff && nelem < (int64_t)0x7ffffff0, "too many targets / elements");
    MM_REQUIRE(ncomp < (1 << 20), "ncomp too large");
    MM_REQUIRE(gll_points_d && (npoints == 0 || points_d), "null array");
    MM_REQUIRE(ncomp == 0 || npoints == 0 || (fields_d && out_d), "null array");
    MM_REQUIRE((elem_out_d == nullptr) == (coeffs_out_d == nullptr), "element and coefficient outputs come together");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    if (npoints == 0) return 0;
    const int P = dim == 3 ? (order + 1) * (order + 1) * (order + 1) : (order + 1) * (order + 1);

    double *cen = nullptr;
    int rc = mm_buffer_get(ctx, MM_BUF_CENTROID, (size_t)nelem * dim * sizeof(double), (void **)&cen);
    if (rc != MM_OK) return rc;
    mm_stage_begin(ctx, MM_STAGE_CENTROID);
    {
        const dim3 g((unsigned)((nelem + 255) / 256)), b(256);
        if (dim == 3) hipLaunchKernelGGL((centroid_nodal_kernel<3>), g, b, 0, ctx->stream, nelem, P, gll_points_d, cen);
        else hipLaunchKernelGGL((centroid_nodal_kernel<2>), g, b, 0, ctx->stream, nelem, P, gll_points_d, cen);
    }
    mm_stage_end(ctx, MM_STAGE_CENTROID);

    mm_knn_index *ix = nullptr;
    mm_stage_begin(ctx, MM_STAGE_KNN_BUILD);
    rc = mm_knn_build_impl(ctx, cen, nelem, dim, &ix, /*use_context_buffers=*/true, nullptr, 0);
    mm_stage_end(ctx, MM_STAGE_KNN_BUILD);
    if (rc != MM_OK) return rc;

    const bool lazy_on = ctx->lazy_lists && k > kGllLazyK;
    const int kavail = lazy_on ? kGllLazyK : (int)k;
    int *nn = nullptr, *nn_full = nullptr;
    rc = mm_buffer_get(ctx, MM_BUF_NN, (size_t)npoints * kavail * sizeof(int), (void **)&nn);
    if (rc == MM_OK && lazy_on)
        rc = mm_buffer_get(ctx, MM_BUF_NN_FULL, (size_t)npoints * k * sizeof(int), (void **)&nn_full);
    if (rc == MM_OK) {
        mm_stage_begin(ctx, MM_STAGE_KNN_QUERY);
        rc = mm_knn_query_impl(ctx, ix, points_d, npoints, kavail, nn, nullptr, /*idx_is_int32=*/true);
        mm_stage_end(ctx, MM_STAGE_KNN_QUERY);
    }
    if (rc == MM_OK) {
        hipError_t e = hipMemsetAsync(ctx->d_counters, 0, sizeof(i64), ctx->stream);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "memset: %s", hipGetErrorString(e));
            rc = MM_ERR_HIP;
        }
    }
    if (rc == MM_OK) {
        mm_stage_begin(ctx, MM_STAGE_LOCATE);
        GllEmit em = {(i64 *)elem_out_d, coeffs_out_d, ncomp > 0 ? fields_d : nullptr, ncomp > 0 ? out_d : nullptr,
                      (int)ncomp};
        // values without the operator: the locate kernels leave {element, xi}, the sums are formed afterwards
        static const bool defer_on = !(getenv("MM_GLL_DEFER") && atoi(getenv("MM_GLL_DEFER")) == 0);
        const bool defer = defer_on && em.out && !em.coeffs;
        if (defer) {
            rc = mm_buffer_get(ctx, MM_BUF_W, (size_t)npoints * dim * sizeof(double), (void **)&em.xi_defer);
            if (rc == MM_OK) rc = mm_buffer_get(ctx, MM_BUF_ENC, (size_t)npoints * sizeof(int), (void **)&em.elem_defer);
        }
        GllLazy lz = {ix, nn_full};
        if (rc == MM_OK)
            rc = locate_gll_run<int>(ctx, order, dim, k, kavail, npoints, nn, gll_points_d, nelem, points_d, tolerance,
                                     snap_to_nearest, em, lazy_on ? &lz : nullptr);
        mm_stage_end(ctx, MM_STAGE_LOCATE);
        if (rc == MM_OK && defer) {
            mm_stage_begin(ctx, MM_STAGE_GATHER);
            const dim3 g((unsigned)((npoints + 255) / 256)), b(256);
#define MM_GLL_VALUES(O, D)                                                                                           \
    if (order == O && dim == D)                                                                                       \
        hipLaunchKernelGGL((gll_values_kernel<O, D>), g, b, 0, ctx->stream, (i64)npoints, (const int *)em.elem_defer,  \
                           (const double *)em.xi_defer, fields_d, (i64)nelem, (int)ncomp, out_d);
            MM_GLL_VALUES(1, 2) MM_GLL_VALUES(1, 3) MM_GLL_VALUES(2, 2) MM_GLL_VALUES(2, 3) MM_GLL_VALUES(4, 2) MM_GLL_VALUES(4, 3)
#undef MM_GLL_VALUES
            mm_stage_end(ctx, MM_STAGE_GATHER);
            if (hipGetLastError() != hipSuccess) {
                mm_set_error(MM_ERR_HIP, "gll_values_kernel launch failed");
                rc = MM_ERR_HIP;
            }
        }
    }
    int64_t result = rc;
    if (rc == MM_OK) {
        hipError_t e = hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_HIP, "pipeline: %s", hipGetErrorString(e));
            result = MM_ERR_HIP;
        } else {
            result = ctx->h_counters[0];
        }
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (ix) mm_knn_destroy(nullptr, ix);   // borrowed arrays stay in the context cache
    return result;
}

extern "C" int mm_gather_elem(mm_context *ctx, const double *fields_d, int64_t nelem, int64_t ncomp,
                              const int64_t *elem_d, const double *coeffs_d, int64_t npoints, int64_t P,
                              double *out_d, int out_point_major)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(nelem >= 0 && ncomp >= 0 && npoints >= 0, "negative size");
    MM_REQUIRE(P >= 1 && P <= 128, "P must be in 1..128");
    MM_REQUIRE(ncomp < (1 << 20), "ncomp too large");
    MM_REQUIRE(npoints == 0 || ncomp == 0 || (fields_d && elem_d && coeffs_d && out_d), "null array");
    MM_REQUIRE(npoints == 0 || ncomp == 0 || nelem >= 1, "empty source mesh");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    if (npoints == 0 || ncomp == 0) return MM_OK;
    const i64 grid = (npoints * 8 + 255) / 256;
    MM_REQUIRE(grid < (i64)0x7fffffff, "too many targets for one launch");
    mm_stage_begin(ctx, MM_STAGE_GATHER);
    if (out_point_major)
        hipLaunchKernelGGL((gather_elem_kernel<true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, fields_d, nelem,
                           (int)ncomp, (const i64 *)elem_d, coeffs_d, npoints, (int)P, out_d);
    else
        hipLaunchKernelGGL((gather_elem_kernel<false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, fields_d,
                           nelem, (int)ncomp, (const i64 *)elem_d, coeffs_d, npoints, (int)P, out_d);
    mm_stage_end(ctx, MM_STAGE_GATHER);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

// Variant 1 of the GLL acceptance loop (bounding-box pre-test; reference interpolator.py:1409-1473).
extern "C" int64_t mm_locate_gll_bbox(mm_context *ctx, int order, int dim, int64_t k, int64_t npoints,
                                      const int64_t *nn_d, const double *gll_points_d, int64_t nelem,
                                      const double *points_d, int64_t *elem_d, double *coeffs_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(order == 1 || order == 2 || order == 4, "order must be 1, 2 or 4");
    MM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    MM_REQUIRE(k >= 0 && npoints >= 0 && nelem >= 0, "negative size");
    MM_REQUIRE(k <= MM_KNN_MAX_K, "nelem_to_search must be <= MM_KNN_MAX_K");
    MM_REQUIRE(npoints == 0 || (elem_d && coeffs_d && points_d), "null array");
    MM_REQUIRE(npoints == 0 || k == 0 || (nn_d && gll_points_d), "null array");
    MM_REQUIRE(npoints < (int64_t)0x7fffffff && nelem < (int64_t)0x7ffffff0, "too many targets / elements");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    MM_HIP_CHECK(hipMemsetAsync(ctx->d_counters, 0, sizeof(i64), ctx->stream));
    if (npoints > 0) {
        mm_stage_begin(ctx, MM_STAGE_LOCATE);
        const i64 nbins = nelem + 1;
        const i64 ntiles = (nbins + 1023) / 1024;
        int rc = mm_scratch_begin(ctx, mm_round256((size_t)(nelem > 0 ? nelem : 1) * 9 * sizeof(double)) +
                                           mm_round256((size_t)npoints * sizeof(int2)) +
                                           mm_round256((size_t)npoints * sizeof(int)) +
                                           2 * mm_round256((size_t)(nbins + 1) * sizeof(int)) +
                                           mm_round256((size_t)ntiles * sizeof(int)) + 4096);
        if (rc != MM_OK) return rc;
        double *boxes = (double *)mm_scratch_take(ctx, (size_t)(nelem > 0 ? nelem : 1) * 9 * sizeof(double));
        MM_REQUIRE(boxes != nullptr, "scratch carve failed");
        const int *visit = nullptr;
        if (k > 0 && nelem > 0) {
            int2 *key_rank = (int2 *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int2));
            int *ord = (int *)mm_scratch_take(ctx, (size_t)npoints * sizeof(int));
            int *counts = (int *)mm_scratch_take(ctx, (size_t)(nbins + 1) * sizeof(int));
            int *start = (int *)mm_scratch_take(ctx, (size_t)(nbins + 1) * sizeof(int));
            int *tile_sums = (int *)mm_scratch_take(ctx, (size_t)ntiles * sizeof(int));
            MM_REQUIRE(key_rank && ord && counts && start && tile_sums, "scratch carve failed");
            MM_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)(nbins + 1) * sizeof(int), ctx->stream));
            const unsigned gp = (unsigned)((npoints + 255) / 256);
            hipLaunchKernelGGL(gll_key_kernel, dim3(gp), dim3(256), 0, ctx->stream, k, npoints, (const i64 *)nn_d, nelem,
                               key_rank, counts);
            rc = mm_exclusive_scan_int(ctx, counts, nbins, start, tile_sums);
            if (rc != MM_OK) return rc;
            hipLaunchKernelGGL(gll_order_kernel, dim3(gp), dim3(256), 0, ctx->stream, npoints, key_rank, start, ord);
            visit = ord;
        }
        unsigned long long *nh = (unsigned long long *)ctx->d_counters;
        const unsigned ge = (unsigned)((nelem + 255) / 256), gt = (unsigned)((npoints + 63) / 64);
#define MM_GLL_V1_CASE(O, D)                                                                                          \
    if (order == O && dim == D) {                                                                                     \
        if (nelem > 0)                                                                                                \
            hipLaunchKernelGGL((gll_box_kernel<O, D>), dim3(ge), dim3(256), 0, ctx->stream, nelem, gll_points_d, boxes); \
        hipLaunchKernelGGL((locate_gll_v1_kernel<O, D>), dim3(gt), dim3(64), 0, ctx->stream, k, npoints,             \
                           (const i64 *)nn_d, gll_points_d, nelem, boxes, points_d, (i64 *)elem_d, coeffs_d, nh, visit); \
    }
        MM_GLL_V1_CASE(1, 2) MM_GLL_V1_CASE(1, 3) MM_GLL_V1_CASE(2, 2) MM_GLL_V1_CASE(2, 3) MM_GLL_V1_CASE(4, 2)
        MM_GLL_V1_CASE(4, 3)
#undef MM_GLL_V1_CASE
        mm_stage_end(ctx, MM_STAGE_LOCATE);
        MM_HIP_CHECK(hipGetLastError());
    }
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ctx->h_counters[0];
}
