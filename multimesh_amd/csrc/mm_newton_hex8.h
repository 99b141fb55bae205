// A6/A7 -- the hex8 Newton inversion of reference multi_mesh/src/trilinearinterpolator.c:199-212 (forward map),
// :214-257 / :320-359 (Jacobian, cofactor inverse) and :260-305 (the iteration), in a form that issues fewer fp64
// instructions than the reference's expressions and still produces the SAME iterates bit for bit.
//
// The file is plain C++ (no HIP types): the kernels include it as device code, tests/test_newton_host.py compiles it
// for the host with g++ and compares every iterate with the CPU oracle on millions of random elements.
//
// What makes a cheaper form legal.  fp64 multiplication by a power of two is exact, so it commutes with every
// rounding (fl(2^k a) = 2^k fl(a), fl(2^k a + 2^k b) = 2^k fl(a + b), fl(1 / (2^k a)) = 2^-k fl(1 / a)) as long as
// nothing overflows or falls into the subnormal range -- coordinates would have to be beyond 1e100 or the element
// smaller than 1e-300 of them, where the reference's own iteration is meaningless.  Two uses:
//   * The reference forms the shape-function derivatives as 0.125 * sign * f * g and sums derivative * corner.  Here
//     the Jacobian is accumulated WITHOUT the factor 0.125 (M = 8 m, every partial sum the reference's times 8), the
//     cofactors are then 64x, the determinant 512x, its reciprocal 1/512x, the inverse 1/8x and the update 1/8x the
//     reference's; the new iterate is fma(8, u/8, xi) = fl(xi + u): the product inside the fused multiply-add is
//     exact, so it rounds once, exactly where the reference's addition does.  12 products f * g per trip instead of
//     the 6 scalings + 18 products the compiler makes of the reference's expression.
//   * The first trip of a fresh solve sits at xi = 0, where every factor is 0.5, 1 or +-0.125: its forward map is 7
//     additions and 11 fused multiply-adds with the exact multiplier +-0.5 per axis (instead of 7 + 7 + 11) and its
//     Jacobian is 72 additions of +-corner (instead of 24 + 72 + 72): 190 instead of 310 instructions.
// Signs ride on the operands (a - t is fl(a + (-t))), so the order of the reference's sums -- node order, from 0 --
// is kept term by term.  Built with -ffp-contract=off: the only fused operations are the ones written here by name.
#pragma once

#if defined(__HIPCC__)
#define MM_HD __host__ __device__ __forceinline__
#else
#define MM_HD inline
#endif

// corner signs of trilinearinterpolator.c:8-10
#define MM_R(n) ((n) == 2 || (n) == 3 || (n) == 5 || (n) == 6 ? 1.0 : -1.0)
#define MM_S(n) ((n) == 1 || (n) == 2 || (n) == 6 || (n) == 7 ? 1.0 : -1.0)
#define MM_T(n) ((n) >= 4 ? 1.0 : -1.0)

// Forward map of one axis (trilinearinterpolator.c:199-212).  The reference's factors are hr = 0.5 * (xi_0 + 1) etc.;
// fr = xi_0 + 1 (the Jacobian needs it anyway) is passed instead and every partial is carried at twice its value:
// e03 = hr * a03 = fl(fr * a03) / 2 exactly, and "x -/+ e03" becomes a fused multiply-add with the exact product
// -/+0.5 * (2 e03) -- the same rounded quantity at every step of the reference's single expression.
MM_HD double map_axis(const double (&v)[8], double fr, double fs, double ft)
{
    const double e03 = fr * (-v[0] + v[3]);   // 2 x the reference's partial, like the next six
    const double e12 = fr * (-v[1] + v[2]);
    const double e45 = fr * (-v[4] + v[5]);
    const double e76 = fr * (v[6] - v[7]);
    const double bottom_s = fs * __builtin_fma(0.5, e12, __builtin_fma(-0.5, e03, -v[0] + v[1]));
    const double top_s = fs * __builtin_fma(0.5, e76, __builtin_fma(-0.5, e45, -v[4] + v[7]));
    const double along_t = ft * __builtin_fma(0.5, top_s, __builtin_fma(-0.5, bottom_s, __builtin_fma(0.5, e45, __builtin_fma(-0.5, e03, -v[0] + v[4]))));
    return __builtin_fma(0.5, along_t, __builtin_fma(0.5, bottom_s, __builtin_fma(0.5, e03, v[0])));
}

// The same map at xi = 0 (hr = hs = ht = 0.5 exactly): e03 = a03 / 2 etc. are exact halvings, so every
// "x -/+ e" of the expression above is a fused multiply-add with the exact product -/+0.5 * a.
MM_HD double map_axis_centre(const double (&v)[8])
{
    const double a03 = -v[0] + v[3];
    const double a12 = -v[1] + v[2];
    const double a45 = -v[4] + v[5];
    const double a76 = v[6] - v[7];
    const double b = __builtin_fma(0.5, a12, __builtin_fma(-0.5, a03, -v[0] + v[1]));    // 2 * bottom_s
    const double t = __builtin_fma(0.5, a76, __builtin_fma(-0.5, a45, -v[4] + v[7]));    // 2 * top_s
    const double l = __builtin_fma(0.5, t, __builtin_fma(-0.5, b, __builtin_fma(0.5, a45, __builtin_fma(-0.5, a03, -v[0] + v[4]))));   // 2 * along_t
    return __builtin_fma(0.5, l, __builtin_fma(0.5, b, __builtin_fma(0.5, a03, v[0])));
}

// One Newton update from the residual r and M = 8 x the reference's Jacobian m[q][j] = sum_n dN_n/dxi_q * corner_n[j]
// (cofactor inverse trilinearinterpolator.c:320-359, update = (J^-1)^T * residual :362-375, each row summed from 0).
MM_HD void newton_update(const double (&M)[3][3], double r0, double r1, double r2, double (&xi)[3])
{
    // the determinant's three cofactors are the first column of the adjugate: the reference writes the middle one as
    // (m10 m22 - m12 m20) there and as (m12 m20 - m10 m22) here -- exact negatives of each other, so
    // "- m01 * (m10 m22 - m12 m20)" is "+ m01 * c10" with the same roundings
    const double c00 = M[1][1] * M[2][2] - M[2][1] * M[1][2];
    const double c10 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
    const double c20 = M[1][0] * M[2][1] - M[2][0] * M[1][1];
    const double det = M[0][0] * c00 + M[0][1] * c10 + M[0][2] * c20;
    const double rdet = 1 / det;
    const double i00 = c00 * rdet;
    const double i01 = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * rdet;
    const double i02 = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * rdet;
    const double i10 = c10 * rdet;
    const double i11 = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * rdet;
    const double i12 = (M[1][0] * M[0][2] - M[0][0] * M[1][2]) * rdet;
    const double i20 = c20 * rdet;
    const double i21 = (M[2][0] * M[0][1] - M[0][0] * M[2][1]) * rdet;
    const double i22 = (M[0][0] * M[1][1] - M[1][0] * M[0][1]) * rdet;
    const double u0 = ((0. + i00 * r0) + i10 * r1) + i20 * r2;   // the reference's update / 8
    const double u1 = ((0. + i01 * r0) + i11 * r1) + i21 * r2;
    const double u2 = ((0. + i02 * r0) + i12 * r1) + i22 * r2;
    xi[0] = __builtin_fma(8.0, u0, xi[0]);
    xi[1] = __builtin_fma(8.0, u1, xi[1]);
    xi[2] = __builtin_fma(8.0, u2, xi[2]);
}

// Newton inversion (trilinearinterpolator.c:260-305).  x/y/z hold the corner coordinates per
// axis.  Returns true when converged; xi receives the last iterate either way.
// first_it > 0 CONTINUES a solve: xi holds the iterate after first_it updates (the iteration is a deterministic map
// of the iterate, so running trips [0, a) and later [a, b) gives the iterates of [0, b)); trips first_it .. max_it - 1.
// POLISH (the GLL path's start, mm_locate_gll.hip): the trip that finds the residual converged still applies its update
// before it returns -- one more hex8 trip (~300 instructions) that takes the iterate from the reference's 1e-8 of an
// element to rounding, which saves a step of the 125-node map (~1.2 k) behind it.  The hex8 path never polishes: its
// iterates are the reference's.
template <bool POLISH = false>
MM_HD bool newton_hex8(const double px, const double py, const double pz, const double (&x)[8],
                       const double (&y)[8], const double (&z)[8], double (&xi)[3], const int max_it = 50,
                       const int first_it = 0)
{
    const double sx = __builtin_fabs(x[1] - x[0]);
    const double sy = __builtin_fabs(y[1] - y[0]);
    const double sz = __builtin_fabs(z[1] - z[0]);
    const double sxy = sx > sy ? sx : sy;
    const double scale = sz > sxy ? sz : sxy;
    const double tol = 1e-8 * scale;
    int it = first_it;
    if (first_it == 0) {
        xi[0] = 0.;
        xi[1] = 0.;
        xi[2] = 0.;
        if (max_it <= 0) return false;
        // trip 0, at the element's centre
        const double r0 = px - map_axis_centre(x);
        const double r1 = py - map_axis_centre(y);
        const double r2 = pz - map_axis_centre(z);
        const bool done0 = __builtin_fabs(r0) < tol && __builtin_fabs(r1) < tol;  // z is never tested (reference quirk)
        if (done0 && !POLISH) return true;
        double M[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            M[q][0] = 0.;
            M[q][1] = 0.;
            M[q][2] = 0.;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            // 8 * dN_n/dxi_q = R, S, T of the node: "+ x" or "- x"
            M[0][0] = MM_R(n) > 0 ? M[0][0] + x[n] : M[0][0] - x[n];
            M[0][1] = MM_R(n) > 0 ? M[0][1] + y[n] : M[0][1] - y[n];
            M[0][2] = MM_R(n) > 0 ? M[0][2] + z[n] : M[0][2] - z[n];
            M[1][0] = MM_S(n) > 0 ? M[1][0] + x[n] : M[1][0] - x[n];
            M[1][1] = MM_S(n) > 0 ? M[1][1] + y[n] : M[1][1] - y[n];
            M[1][2] = MM_S(n) > 0 ? M[1][2] + z[n] : M[1][2] - z[n];
            M[2][0] = MM_T(n) > 0 ? M[2][0] + x[n] : M[2][0] - x[n];
            M[2][1] = MM_T(n) > 0 ? M[2][1] + y[n] : M[2][1] - y[n];
            M[2][2] = MM_T(n) > 0 ? M[2][2] + z[n] : M[2][2] - z[n];
        }
        newton_update(M, r0, r1, r2, xi);
        if (POLISH && done0) return true;
        it = 1;
    }
    for (; it < max_it; ++it) {
        // the reference's factors xi * (+-1) + 1 (two values per axis)
        const double fr[2] = {-xi[0] + 1, xi[0] + 1};
        const double fs[2] = {-xi[1] + 1, xi[1] + 1};
        const double ft[2] = {-xi[2] + 1, xi[2] + 1};
        const double r0 = px - map_axis(x, fr[1], fs[1], ft[1]);
        const double r1 = py - map_axis(y, fr[1], fs[1], ft[1]);
        const double r2 = pz - map_axis(z, fr[1], fs[1], ft[1]);
        const bool done = __builtin_fabs(r0) < tol && __builtin_fabs(r1) < tol;  // z is never tested (reference quirk)
        if (done && !POLISH) return true;
        // ... and their pairwise products: dN_n/dxi_0 = 0.125 R_n fs ft is an exact scaling of fl(fs ft), the same for
        // the other two
        double gst[2][2], grt[2][2], grs[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                gst[a][b] = fs[a] * ft[b];
                grt[a][b] = fr[a] * ft[b];
                grs[a][b] = fr[a] * fs[b];
            }
        double M[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            M[q][0] = 0.;
            M[q][1] = 0.;
            M[q][2] = 0.;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int ir = MM_R(n) > 0, is = MM_S(n) > 0, it_ = MM_T(n) > 0;
            const double d0 = gst[is][it_], d1 = grt[ir][it_], d2 = grs[ir][is];   // |8 dN_n/dxi_q|
            M[0][0] = ir ? M[0][0] + d0 * x[n] : M[0][0] - d0 * x[n];
            M[0][1] = ir ? M[0][1] + d0 * y[n] : M[0][1] - d0 * y[n];
            M[0][2] = ir ? M[0][2] + d0 * z[n] : M[0][2] - d0 * z[n];
            M[1][0] = is ? M[1][0] + d1 * x[n] : M[1][0] - d1 * x[n];
            M[1][1] = is ? M[1][1] + d1 * y[n] : M[1][1] - d1 * y[n];
            M[1][2] = is ? M[1][2] + d1 * z[n] : M[1][2] - d1 * z[n];
            M[2][0] = it_ ? M[2][0] + d2 * x[n] : M[2][0] - d2 * x[n];
            M[2][1] = it_ ? M[2][1] + d2 * y[n] : M[2][1] - d2 * y[n];
            M[2][2] = it_ ? M[2][2] + d2 * z[n] : M[2][2] - d2 * z[n];
        }
        newton_update(M, r0, r1, r2, xi);
        if (POLISH && done) return true;
    }
    return false;
}
