// Host build of multimesh_amd/csrc/mm_newton_hex8.h for tests/test_newton_host.py (test infrastructure; the product
// compiles the same header as device code).  g++ -O2 -mfma -ffp-contract=off: the only fused operations are the
// __builtin_fma calls the header spells out.
#include <cstdint>
#include <cstring>

#include "../../multimesh_amd/csrc/mm_newton_hex8.h"

extern "C" {

// one solve, corners as the reference passes them (vtx[8][3]); trips first_it .. max_it - 1
int nh_newton(const double *pnt, const double *vtx, double *xi, int max_it, int first_it)
{
    double x[8], y[8], z[8];
    for (int n = 0; n < 8; ++n) {
        x[n] = vtx[n * 3 + 0];
        y[n] = vtx[n * 3 + 1];
        z[n] = vtx[n * 3 + 2];
    }
    double q[3] = {xi[0], xi[1], xi[2]};
    const bool ok = newton_hex8(pnt[0], pnt[1], pnt[2], x, y, z, q, max_it, first_it);
    xi[0] = q[0];
    xi[1] = q[1];
    xi[2] = q[2];
    return ok ? 1 : 0;
}

// the corner solve the GLL path starts from (newton_hex8_start: the polynomial form, at most cap trips)
int nh_start(const double *pnt, const double *vtx, double *xi, int cap)
{
    double x[8], y[8], z[8];
    for (int n = 0; n < 8; ++n) {
        x[n] = vtx[n * 3 + 0];
        y[n] = vtx[n * 3 + 1];
        z[n] = vtx[n * 3 + 2];
    }
    double q[3] = {0, 0, 0};
    newton_hex8_start(pnt[0], pnt[1], pnt[2], x, y, z, q, cap);
    xi[0] = q[0];
    xi[1] = q[1];
    xi[2] = q[2];
    return 1;
}

// other(pnt, vtx, xi, iters) -> converged: the oracle's mmo_hex8_newton (iters may be null), or with no_iters != 0
// the compiled reference's inverseCoordinateTransform(pnt, vtx, xi).
typedef int (*other4_t)(const double *, const double *, double *, int *);
typedef int (*other3_t)(const double *, const double *, double *);

static bool same_bits(const double *a, const double *b)
{
    for (int q = 0; q < 3; ++q) {
        if (a[q] != a[q] && b[q] != b[q]) continue;   // NaN on both sides (payloads are not part of the contract)
        if (std::memcmp(a + q, b + q, sizeof(double)) != 0) return false;
    }
    return true;
}

typedef void (*start_t)(const double *, const double *, double *, int);

// n corner solves against other = the oracle's mmo_hex8_newton_start: solves that differ in verdict or iterate
int64_t nh_compare_start(int64_t n, const double *pnts, const double *vtxs, void *other, int cap, int64_t *first_bad)
{
    int64_t bad = 0;
    *first_bad = -1;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = pnts + i * 3, *v = vtxs + i * 24;
        double xo[3] = {0, 0, 0}, xm[3] = {0, 0, 0};
        ((start_t)other)(p, v, xo, cap);
        (void)nh_start(p, v, xm, cap);
        if (!same_bits(xo, xm)) {
            if (*first_bad < 0) *first_bad = i;
            ++bad;
        }
    }
    return bad;
}

// n solves; staged != 0 runs this library's solve the way the pass kernel does (caps c1 -> c2 -> 50, every stage
// continuing from the iterate the stage before stopped at).  Returns the number of solves whose verdict or final
// iterate differs from other's; first_bad receives the index of the first one (or -1); converged the count.
int64_t nh_compare(int64_t n, const double *pnts, const double *vtxs, void *other, int no_iters, int staged, int c1,
                   int c2, int64_t *first_bad, int64_t *converged)
{
    int64_t bad = 0, conv = 0;
    *first_bad = -1;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = pnts + i * 3, *v = vtxs + i * 24;
        double xo[3] = {0, 0, 0}, xm[3] = {0, 0, 0};
        const int ok_o = no_iters ? ((other3_t)other)(p, v, xo) : ((other4_t)other)(p, v, xo, nullptr);
        int ok_m;
        if (staged) {
            ok_m = nh_newton(p, v, xm, c1, 0);
            if (!ok_m) ok_m = nh_newton(p, v, xm, c2, c1);
            if (!ok_m) ok_m = nh_newton(p, v, xm, 50, c2);
        } else {
            ok_m = nh_newton(p, v, xm, 50, 0);
        }
        conv += ok_m;
        if ((ok_o != 0) != (ok_m != 0) || !same_bits(xo, xm)) {
            if (*first_bad < 0) *first_bad = i;
            ++bad;
        }
    }
    *converged = conv;
    return bad;
}

// MM_FP_TOL: n solves by newton_hex8_fast (cap trips) against other = the oracle's mmo_hex8_newton (the reference
// iteration, 50 trips).  out[0..2] solves certified accept / certified reject / unsure; out[3] certified verdicts that
// are WRONG (accept where the reference rejects or the other way round: must be 0); out[4] certified solves whose trip
// count differs from the reference's; out[5] unsure solves the reference would have accepted.
// dout[0] largest |xi_fast - xi_ref| / delta over the certified solves, dout[1] largest |xi_fast - xi_ref|, dout[2] the
// largest delta handed out, dout[3] the largest update ratio |u_i| / |u_i-1| among the certified.
void nh_fast_stats(int64_t n, const double *pnts, const double *vtxs, void *other, int cap, int64_t *out, double *dout)
{
    for (int q = 0; q < 6; ++q) out[q] = 0;
    for (int q = 0; q < 4; ++q) dout[q] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = pnts + i * 3, *v = vtxs + i * 24;
        double xo[3] = {0, 0, 0}, xm[3] = {0, 0, 0}, diag[3];
        int iters = 0;
        const int ok_o = ((other4_t)other)(p, v, xo, &iters);
        double worst = 0;
        for (int q = 0; q < 3; ++q) worst = __builtin_fabs(xo[q]) > worst ? __builtin_fabs(xo[q]) : worst;
        const bool accept_ref = ok_o && worst < (1 + 0.025);
        double x[8], y[8], z[8];
        for (int c = 0; c < 8; ++c) {
            x[c] = v[c * 3 + 0];
            y[c] = v[c * 3 + 1];
            z[c] = v[c * 3 + 2];
        }
        const int verdict = newton_hex8_fast(p[0], p[1], p[2], x, y, z, xm, cap, diag);
        if (verdict == MM_FAST_UNSURE) {
            ++out[2];
            if (accept_ref) ++out[5];
            continue;
        }
        ++out[verdict == MM_FAST_ACCEPT ? 0 : 1];
        if ((verdict == MM_FAST_ACCEPT) != accept_ref) ++out[3];
        // (a certified verdict claims convergence at trip diag[1]: the reference's count is iters = that trip + 1)
        if (!ok_o || iters != (int)diag[1] + 1) {
            ++out[4];
            continue;
        }
        double d = 0;
        for (int q = 0; q < 3; ++q) d = __builtin_fabs(xo[q] - xm[q]) > d ? __builtin_fabs(xo[q] - xm[q]) : d;
        if (d / diag[0] > dout[0]) dout[0] = d / diag[0];
        if (d > dout[1]) dout[1] = d;
        if (diag[0] > dout[2]) dout[2] = diag[0];
        if (diag[2] > dout[3]) dout[3] = diag[2];
    }
}

// the MM_FP_TOL weights against other = the oracle's mmo_hex8_weights(xi, w): largest absolute difference
double nh_fast_weights(int64_t n, const double *xis, void (*other)(const double *, double *))
{
    double worst = 0;
    for (int64_t i = 0; i < n; ++i) {
        double xi[3] = {xis[i * 3], xis[i * 3 + 1], xis[i * 3 + 2]}, a[8], b[8];
        weights_hex8_fast(xi, a);
        other(xi, b);
        for (int q = 0; q < 8; ++q) worst = __builtin_fabs(a[q] - b[q]) > worst ? __builtin_fabs(a[q] - b[q]) : worst;
    }
    return worst;
}
}
