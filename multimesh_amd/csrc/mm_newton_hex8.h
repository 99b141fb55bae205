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

// =====================================================================================================================
// MM_FP_TOL (mm_set_fp_mode): the same DECISIONS as the iteration above from a quarter of its instructions.
//
// The reference's operation order costs ~300 fp64 instructions per trip.  north_star asks for element indices bit-exact
// and fields "within a stated relative tolerance", so a second solve is allowed to round differently as long as it
// reaches the reference's verdict -- converged or not, max|xi| on which side of 1.025 -- with certainty, and says so
// when it cannot.  newton_hex8_fast:
//   * writes the trilinear map as the polynomial c0 + r cR + s cS + t cT + rs cRS + rt cRT + st cST + rst cRST (the
//     coefficients, carried at 8x, are a Walsh-Hadamard butterfly of the corners: 24 additions per axis and solve),
//     so that one trip's residual AND Jacobian are 11 fused multiply-adds per axis (at xi = 0 they are c0 and
//     cR / cS / cT themselves), and solves the 3x3 system by cross products (Cramer) with a refined reciprocal:
//     ~85 instructions per trip;
//   * follows the reference's control flow (start at 0, residual test on x and y only, update, cap);
//   * returns MM_FAST_ACCEPT / MM_FAST_REJECT only when every test the reference makes on the way has MARGIN:
//       - the two iterations start from the same point and differ by rounding alone, which one update turns into at
//         most E ~ eps * |coordinates| * |J^-1| in xi (a few 1e-13 on the metric meshes).  Newton's map has the
//         derivative -J^-1 DJ u: where the updates u shrink at least two-fold per trip it is a contraction and the
//         difference stays at the level of E trip after trip; that is checked (trip i >= 1: 2 |u_i| <= |u_i-1|);
//       - every residual test must then fall outside tol -+ rho and the final max|xi| outside 1.025 -+ delta, with
//         delta = 256 eps |v| |J^-1| and rho = 64 eps |v| + |J| delta taken from the element at xi = 0 -- two to
//         three orders above the differences measured (tests/test_newton_host.py runs millions of solves against
//         the reference iteration and reports the largest difference in units of delta);
//     anything else -- a test inside its band, updates that do not shrink, no convergence inside the cap, a
//     non-finite number, an element so ill-conditioned that delta > 1e-6 -- returns MM_FAST_UNSURE and the caller
//     repeats the solve with newton_hex8 (locate_pass_kernel<..., FAST> hands such targets to the exact kernel).
// With MM_FAST_ACCEPT xi is the reference's final iterate to within delta (same trip count).
// =====================================================================================================================
#define MM_FAST_REJECT 0
#define MM_FAST_ACCEPT 1
#define MM_FAST_UNSURE 2

constexpr double kFastEps = 2.220446049250313e-16;
constexpr double kFastCxi = 256.0;     // delta = kFastCxi * eps * |v|_8 * |(8J)^-1|
constexpr double kFastCres = 64.0;     // rho_8 = kFastCres * eps * |v|_8 + 3 |8J|_max * delta
constexpr double kFastDeltaMax = 1e-6; // elements worse conditioned than this are not certified at all

// 1 / d to about 2^-26 (what v_rcp_f64 delivers; the host build cuts an exact quotient down to that, so that the
// host tests exercise the refinement below rather than a better seed than the device has)
MM_HD double mm_rcp_seed(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);
#else
    double y = 1.0 / d;
    unsigned long long b;
    __builtin_memcpy(&b, &y, 8);
    b &= ~((1ull << 26) - 1ull);
    __builtin_memcpy(&y, &b, 8);
    return y;
#endif
}

MM_HD double mm_max3abs(double a, double b, double c)
{
    const double m = __builtin_fmax(__builtin_fabs(a), __builtin_fabs(b));
    return __builtin_fmax(m, __builtin_fabs(c));
}

// 8 x the polynomial coefficients of one axis (corner signs MM_R / MM_S / MM_T): {c0, cR, cS, cT, cRS, cRT, cST, cRST}
MM_HD void hex8_poly_axis(const double (&v)[8], double (&c)[8])
{
    // along r: corner pairs (0,3) (1,2) (4,5) (7,6); along s; along t
    const double s_mm = v[0] + v[3], d_mm = v[3] - v[0];
    const double s_pm = v[1] + v[2], d_pm = v[2] - v[1];
    const double s_mp = v[4] + v[5], d_mp = v[5] - v[4];
    const double s_pp = v[7] + v[6], d_pp = v[6] - v[7];
    const double ss_m = s_mm + s_pm, sd_m = s_pm - s_mm;
    const double ss_p = s_mp + s_pp, sd_p = s_pp - s_mp;
    const double ds_m = d_mm + d_pm, dd_m = d_pm - d_mm;
    const double ds_p = d_mp + d_pp, dd_p = d_pp - d_mp;
    c[0] = ss_m + ss_p;
    c[3] = ss_p - ss_m;
    c[2] = sd_m + sd_p;
    c[6] = sd_p - sd_m;
    c[1] = ds_m + ds_p;
    c[5] = ds_p - ds_m;
    c[4] = dd_m + dd_p;
    c[7] = dd_p - dd_m;
}

// Cramer's rule for sum_q a_q u_q = res (a_r, a_s, a_t: the rows of 8 J as vectors in x, y, z): the three cross products
// nr = as x at, ns = at x ar, nt = ar x as and 1 / det, det = ar . nr; then u_q = (res . n_q) / det.
MM_HD void fast_invert(const double (&ar)[3], const double (&as)[3], const double (&at)[3], double (&nr)[3],
                       double (&ns)[3], double (&nt)[3], double &rdet)
{
#define MM_CROSS(o, a, b)                                      \
    o[0] = __builtin_fma(a[1], b[2], -(a[2] * b[1]));          \
    o[1] = __builtin_fma(a[2], b[0], -(a[0] * b[2]));          \
    o[2] = __builtin_fma(a[0], b[1], -(a[1] * b[0]));
    MM_CROSS(nr, as, at)
    MM_CROSS(ns, at, ar)
    MM_CROSS(nt, ar, as)
#undef MM_CROSS
    const double det = __builtin_fma(ar[2], nr[2], __builtin_fma(ar[1], nr[1], ar[0] * nr[0]));
    double y = mm_rcp_seed(det);
    y = __builtin_fma(y, __builtin_fma(-det, y, 1.0), y);
    rdet = __builtin_fma(y, __builtin_fma(-det, y, 1.0), y);
}

MM_HD double fast_dot(const double (&a)[3], const double (&b)[3])
{
    return __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], a[0] * b[0]));
}

// one axis of the map at (r, s, t): its three gradients (8x) and 8 x the residual, 11 fused multiply-adds
MM_HD void fast_axis(const double (&c)[8], double q0, double r, double s, double t, double &gr, double &gs, double &gt,
                     double &res)
{
    const double A = __builtin_fma(t, c[7], c[4]);
    const double D = __builtin_fma(t, c[6], c[2]);
    gr = __builtin_fma(s, A, __builtin_fma(t, c[5], c[1]));
    gs = __builtin_fma(r, A, D);
    gt = __builtin_fma(s, __builtin_fma(r, c[7], c[6]), __builtin_fma(r, c[5], c[3]));
    res = __builtin_fma(-t, c[3], __builtin_fma(-s, D, __builtin_fma(-r, gr, q0)));
}

// diag (nullable, host tests): {delta, the trip that found convergence, largest |u_i| / |u_i-1| seen}
MM_HD int newton_hex8_fast(const double px, const double py, const double pz, const double (&x)[8],
                           const double (&y)[8], const double (&z)[8], double (&xi)[3], const int max_it,
                           double *diag = nullptr)
{
    // the reference's tolerance (trilinearinterpolator.c:278-282), at 8x like everything below
    const double sx = __builtin_fabs(x[1] - x[0]);
    const double sy = __builtin_fabs(y[1] - y[0]);
    const double sz = __builtin_fabs(z[1] - z[0]);
    const double tol8 = 8e-8 * __builtin_fmax(__builtin_fmax(sx, sy), sz);
    double cx[8], cy[8], cz[8];
    hex8_poly_axis(x, cx);
    hex8_poly_axis(y, cy);
    hex8_poly_axis(z, cz);
    if (diag) diag[0] = diag[1] = diag[2] = 0.;
    // Trip 0, at xi = 0: the rows of 8 J and 8 x the residual are the coefficients themselves.
    double ar[3] = {cx[1], cy[1], cz[1]}, as[3] = {cx[2], cy[2], cz[2]}, at[3] = {cx[3], cy[3], cz[3]};
    const double q0[3] = {__builtin_fma(8.0, px, -cx[0]), __builtin_fma(8.0, py, -cy[0]), __builtin_fma(8.0, pz, -cz[0])};
    double res[3] = {q0[0], q0[1], q0[2]};
    double nr[3], ns[3], nt[3], rdet;
    fast_invert(ar, as, at, nr, ns, nt, rdet);
    // the margins, from the element at its centre: |(8J)^-1| <= 3 max|cofactor| / |det| (row sums), |v|_8 = 8 x the
    // largest coordinate around (corner sums, the point, the element's extent)
    double cof = mm_max3abs(nr[0], nr[1], nr[2]);
    cof = __builtin_fmax(cof, mm_max3abs(ns[0], ns[1], ns[2]));
    cof = __builtin_fmax(cof, mm_max3abs(nt[0], nt[1], nt[2]));
    double jmax = mm_max3abs(ar[0], ar[1], ar[2]);
    jmax = __builtin_fmax(jmax, mm_max3abs(as[0], as[1], as[2]));
    jmax = __builtin_fmax(jmax, mm_max3abs(at[0], at[1], at[2]));
    const double v8 = __builtin_fma(4.0, jmax, __builtin_fmax(mm_max3abs(cx[0], cy[0], cz[0]), 8.0 * mm_max3abs(px, py, pz)));
    const double delta = (kFastCxi * kFastEps) * v8 * (3.0 * cof * __builtin_fabs(rdet));
    const double rho8 = __builtin_fma(3.0 * jmax, delta, (kFastCres * kFastEps) * v8);
    const double lo8 = tol8 - rho8, hi8 = tol8 + rho8;
    if (diag) diag[0] = delta;
    // (one exit: the structuriser turns early returns from inside the loop into a web of register copies)
    int verdict = MM_FAST_UNSURE;
    double r = 0., s = 0., t = 0., m_prev = 0.;
    int it = 0;
    bool go = delta < kFastDeltaMax && lo8 > 0.;   // (a NaN gives false)
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    while (go) {
        // the reference's stop test (x and y only), with its band
        const double a0 = __builtin_fabs(res[0]), a1 = __builtin_fabs(res[1]);
        if (a0 < lo8 && a1 < lo8) {
            if (diag) diag[1] = it;
            const double worst = mm_max3abs(r, s, t);
            verdict = worst < (1 + 0.025) - delta ? MM_FAST_ACCEPT : (worst > (1 + 0.025) + delta ? MM_FAST_REJECT : MM_FAST_UNSURE);
            break;
        }
        // inside the band (or not a number); or no verdict inside the cap: the reference may run to 50
        if (!(a0 > hi8 || a1 > hi8) || it + 1 >= max_it) break;
        if (it > 0) fast_invert(ar, as, at, nr, ns, nt, rdet);
        const double dr = fast_dot(res, nr), ds = fast_dot(res, ns), dt = fast_dot(res, nt);
        const double m = mm_max3abs(dr, ds, dt) * __builtin_fabs(rdet);   // |u|
        if (diag && it > 0 && m_prev > 0. && m / m_prev > diag[2]) diag[2] = m / m_prev;
        if (it > 0 && !(2.0 * m <= m_prev)) break;   // not (yet) contracting: roundings may grow
        m_prev = m;
        r = __builtin_fma(dr, rdet, r);
        s = __builtin_fma(ds, rdet, s);
        t = __builtin_fma(dt, rdet, t);
        ++it;
        fast_axis(cx, q0[0], r, s, t, ar[0], as[0], at[0], res[0]);
        fast_axis(cy, q0[1], r, s, t, ar[1], as[1], at[1], res[1]);
        fast_axis(cz, q0[2], r, s, t, ar[2], as[2], at[2], res[2]);
    }
    xi[0] = r;
    xi[1] = s;
    xi[2] = t;
    return verdict;
}

// The eight weights in factored form, 0.125 (1 +- r)(1 +- s)(1 +- t) -- MM_FP_TOL only (the exact path keeps the
// reference's expanded polynomials, trilinearinterpolator.c:174-197): within a few ulp of those.
MM_HD void weights_hex8_fast(const double (&xi)[3], double (&w)[8])
{
    const double fr[2] = {__builtin_fma(-0.125, xi[0], 0.125), __builtin_fma(0.125, xi[0], 0.125)};
    const double fs[2] = {1.0 - xi[1], 1.0 + xi[1]};
    const double ft[2] = {1.0 - xi[2], 1.0 + xi[2]};
    const double g[2][2] = {{fs[0] * ft[0], fs[0] * ft[1]}, {fs[1] * ft[0], fs[1] * ft[1]}};
#define MM_W(n) w[n] = fr[MM_R(n) > 0] * g[MM_S(n) > 0][MM_T(n) > 0];
    MM_W(0) MM_W(1) MM_W(2) MM_W(3) MM_W(4) MM_W(5) MM_W(6) MM_W(7)
#undef MM_W
}

// The start of a 3-D GLL inverse transform (mm_locate_gll.hip, round 4): Newton on the element's eight CORNERS' trilinear
// map in the polynomial form above -- at most max_it trips, every trip's update applied, stopped after the trip whose
// update is below 1e-9 (the 125-node iteration behind it polishes) or when an iterate stops being finite.  The GLL path's
// arithmetic is this project's own definition (salvus.fem is absent: parity unpinned), so unlike newton_hex8_fast there is
// nothing to certify: the oracle's mmo_hex8_start restates THESE operations (fma for fma, an exact division where the hex8
// path refines a reciprocal) and tests/test_newton_host.py compares the two bit for bit.  ~80 fp64 instructions per trip
// against the reference-order iteration's ~300, a third of its registers.
MM_HD void newton_hex8_start(const double px, const double py, const double pz, const double (&x)[8], const double (&y)[8],
                             const double (&z)[8], double (&xi)[3], const int max_it)
{
    double cx[8], cy[8], cz[8];
    hex8_poly_axis(x, cx);
    hex8_poly_axis(y, cy);
    hex8_poly_axis(z, cz);
    const double q0[3] = {__builtin_fma(8.0, px, -cx[0]), __builtin_fma(8.0, py, -cy[0]), __builtin_fma(8.0, pz, -cz[0])};
    double ar[3] = {cx[1], cy[1], cz[1]}, as[3] = {cx[2], cy[2], cz[2]}, at[3] = {cx[3], cy[3], cz[3]};
    double res[3] = {q0[0], q0[1], q0[2]};
    double r = 0., s = 0., t = 0.;
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int it = 0; it < max_it; ++it) {
        double nr[3], ns[3], nt[3];
#define MM_CROSS(o, a, b)                                      \
    o[0] = __builtin_fma(a[1], b[2], -(a[2] * b[1]));          \
    o[1] = __builtin_fma(a[2], b[0], -(a[0] * b[2]));          \
    o[2] = __builtin_fma(a[0], b[1], -(a[1] * b[0]));
        MM_CROSS(nr, as, at)
        MM_CROSS(ns, at, ar)
        MM_CROSS(nt, ar, as)
#undef MM_CROSS
        const double det = __builtin_fma(ar[2], nr[2], __builtin_fma(ar[1], nr[1], ar[0] * nr[0]));
        const double rdet = 1.0 / det;
        const double dr = fast_dot(res, nr), ds = fast_dot(res, ns), dt = fast_dot(res, nt);
        const double m = mm_max3abs(dr, ds, dt) * __builtin_fabs(rdet);
        r = __builtin_fma(dr, rdet, r);
        s = __builtin_fma(ds, rdet, s);
        t = __builtin_fma(dt, rdet, t);
        if (!(m >= 1e-9)) break;               // converged (or not a number: the caller looks at the iterate)
        if (!(mm_max3abs(r, s, t) <= 1e3)) break;   // running away: not a usable start
        fast_axis(cx, q0[0], r, s, t, ar[0], as[0], at[0], res[0]);
        fast_axis(cy, q0[1], r, s, t, ar[1], as[1], at[1], res[1]);
        fast_axis(cz, q0[2], r, s, t, ar[2], as[2], at[2], res[2]);
    }
    xi[0] = r;
    xi[1] = s;
    xi[2] = t;
}
