/*
 * mm_oracle.c -- CPU ORACLE for the MultiMesh interpolation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under multimesh_amd/ may include, link,
 * load or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * This is an independent plain-C restatement of the reference algorithm.  Each
 * function cites the reference lines it follows (paths relative to
 * /root/reference/multi_mesh/).  Floating-point expressions keep the
 * reference's association order, because parity for the weights is judged
 * bit-for-bit; compile with -ffp-contract=off (see oracle/Makefile).
 *
 * Pinning: tests/test_oracle_pinned.py compares every hex8 function below with
 * the reference's own C translation units compiled into oracle/_ref/ (when
 * /root/reference is present) and with the committed fixtures in
 * tests/golden/ that were generated from that build.
 *
 * The GLL section at the end (A10) is PARITY UNPINNED: its numerics live in the
 * proprietary salvus.fem package the reference imports, which is absent and has
 * no test or fixture in the reference; only its control flow follows the
 * reference, and it is pinned by analytic properties (tests/test_gll_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

/* ------------------------------------------------------------------ */
/* A1: element centroid.  Follows src/centroid.c:15-23: per axis, sum  */
/* the element's nodes in connectivity order, then divide by the node  */
/* count (a division, not a multiply by the reciprocal).               */
/* ------------------------------------------------------------------ */
void mmo_centroid(i64 ndim, i64 nelem, i64 nper, const i64 *conn,
                  const double *points, double *out)
{
    for (i64 e = 0; e < nelem; ++e) {
        const i64 *row = conn + e * nper;
        for (i64 a = 0; a < ndim; ++a) {
            double acc = 0.;
            for (i64 p = 0; p < nper; ++p)
                acc = acc + points[row[p] * ndim + a];
            out[e * ndim + a] = acc / nper;
        }
    }
}

/* ------------------------------------------------------------------ */
/* A7: forward trilinear map of one coordinate axis                    */
/* (src/trilinearinterpolator.c:199-212).  Node k sits at the corner   */
/* (R,S,T)[k] of src/trilinearinterpolator.c:8-10.  Written with named */
/* partial results; each partial is the same rounded quantity the      */
/* reference's single expression produces, so the result is identical. */
/* ------------------------------------------------------------------ */
static double hex8_map_axis(const double v[8], double r, double s, double t)
{
    const double hr = 0.5 * (r + 1.0);
    const double hs = 0.5 * (s + 1.0);
    const double ht = 0.5 * (t + 1.0);
    const double e03 = hr * (-v[0] + v[3]);
    const double e12 = hr * (-v[1] + v[2]);
    const double e45 = hr * (-v[4] + v[5]);
    const double e76 = hr * (v[6] - v[7]);
    const double bottom_s = hs * (((-v[0] + v[1]) - e03) + e12);
    const double top_s = hs * (((-v[4] + v[7]) - e45) + e76);
    const double along_t = ht * (((((-v[0] + v[4]) - e03) + e45) - bottom_s) + top_s);
    return ((v[0] + e03) + bottom_s) + along_t;
}

static const double kR[8] = {-1, -1, +1, +1, -1, +1, +1, -1};
static const double kS[8] = {-1, +1, +1, -1, -1, -1, +1, +1};
static const double kT[8] = {-1, -1, -1, -1, +1, +1, +1, +1};

/* Jacobian J[q][j] = sum_i dN_i/dxi_q * x_i[j] and its inverse via cofactors
 * (src/trilinearinterpolator.c:214-257, :320-341, :344-359). */
static void hex8_inverse_jacobian(const double xi[3], const double vtx[8][3],
                                  double inv[3][3])
{
    double dn[3][8];
    for (int n = 0; n < 8; ++n) {
        dn[0][n] = 0.125 * kR[n] * (xi[1] * kS[n] + 1) * (xi[2] * kT[n] + 1);
        dn[1][n] = 0.125 * kS[n] * (xi[0] * kR[n] + 1) * (xi[2] * kT[n] + 1);
        dn[2][n] = 0.125 * kT[n] * (xi[0] * kR[n] + 1) * (xi[1] * kS[n] + 1);
    }
    double m[3][3];
    for (int q = 0; q < 3; ++q)
        for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int i = 0; i < 8; ++i)
                acc = acc + dn[q][i] * vtx[i][j];
            m[q][j] = acc;
        }
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) -
                       m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    const double rdet = 1 / det;
    inv[0][0] = (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * rdet;
    inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * rdet;
    inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * rdet;
    inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * rdet;
    inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * rdet;
    inv[1][2] = (m[1][0] * m[0][2] - m[0][0] * m[1][2]) * rdet;
    inv[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * rdet;
    inv[2][1] = (m[2][0] * m[0][1] - m[0][0] * m[2][1]) * rdet;
    inv[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * rdet;
}

/* ------------------------------------------------------------------ */
/* A6: Newton inversion (src/trilinearinterpolator.c:260-305).         */
/* xi0 = 0; tol = 1e-8 * max|v1-v0| over the axes; at most 50 steps;   */
/* the convergence test looks at residual x, y and x again -- z is     */
/* never tested (:290-291), and that quirk is kept on purpose.  The    */
/* update is xi += (J^-1)^T * residual (:295-300).                     */
/* Returns 1 when converged, 0 otherwise; *iters = residual tests done.*/
/* ------------------------------------------------------------------ */
static int hex8_newton_capped(const double pnt[3], const double vtx[8][3], double xi[3], int *iters, int cap, int polish);

int mmo_hex8_newton(const double pnt[3], const double vtx[8][3], double xi[3],
                    int *iters)
{
    return hex8_newton_capped(pnt, vtx, xi, iters, 50, 0);
}

/* The corner solve the GLL section starts from (test hook: tests/test_newton_host.py compares it with the kernels'). */
int mmo_hex8_newton_start(const double pnt[3], const double vtx[8][3], double xi[3], int cap)
{
    return hex8_newton_capped(pnt, vtx, xi, NULL, cap, 1);
}

/* The start of a 3-D GLL inverse transform since round 4 (the GLL section's own arithmetic: salvus.fem is absent, parity
 * unpinned): Newton on the eight corners' trilinear map written as the polynomial c0 + r cR + s cS + t cT + rs cRS + rt cRT +
 * st cST + rst cRST (coefficients at 8x: a Walsh-Hadamard butterfly of the corners), Cramer's rule by cross products,
 * every partial result by the fused multiply-adds spelled out here -- operation for operation what newton_hex8_start of
 * multimesh_amd/csrc/mm_newton_hex8.h does (tests/test_newton_host.py compares the two bit for bit).  At most cap trips,
 * every update applied; stops after an update below 1e-9 or when an iterate leaves [-1e3, 1e3] / is not a number. */
static void hex8_poly_axis(const double v[8], double c[8])
{
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

static void hex8_fast_axis(const double c[8], double q0, double r, double s, double t, double *gr, double *gs, double *gt,
                           double *res)
{
    const double A = fma(t, c[7], c[4]);
    const double D = fma(t, c[6], c[2]);
    *gr = fma(s, A, fma(t, c[5], c[1]));
    *gs = fma(r, A, D);
    *gt = fma(s, fma(r, c[7], c[6]), fma(r, c[5], c[3]));
    *res = fma(-t, c[3], fma(-s, D, fma(-r, *gr, q0)));
}

static void hex8_cross(double o[3], const double a[3], const double b[3])
{
    o[0] = fma(a[1], b[2], -(a[2] * b[1]));
    o[1] = fma(a[2], b[0], -(a[0] * b[2]));
    o[2] = fma(a[0], b[1], -(a[1] * b[0]));
}

static double hex8_dot(const double a[3], const double b[3]) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }

static double max3abs(double a, double b, double c) { return fmax(fmax(fabs(a), fabs(b)), fabs(c)); }

void mmo_hex8_start(const double pnt[3], const double vtx[8][3], double xi[3], int cap)
{
    double col[8], c[3][8];
    for (int a = 0; a < 3; ++a) {
        for (int n = 0; n < 8; ++n) col[n] = vtx[n][a];
        hex8_poly_axis(col, c[a]);
    }
    const double q0[3] = {fma(8.0, pnt[0], -c[0][0]), fma(8.0, pnt[1], -c[1][0]), fma(8.0, pnt[2], -c[2][0])};
    double ar[3] = {c[0][1], c[1][1], c[2][1]}, as[3] = {c[0][2], c[1][2], c[2][2]}, at[3] = {c[0][3], c[1][3], c[2][3]};
    double res[3] = {q0[0], q0[1], q0[2]};
    double r = 0., s = 0., t = 0.;
    for (int it = 0; it < cap; ++it) {
        double nr[3], ns[3], nt[3];
        hex8_cross(nr, as, at);
        hex8_cross(ns, at, ar);
        hex8_cross(nt, ar, as);
        const double det = fma(ar[2], nr[2], fma(ar[1], nr[1], ar[0] * nr[0]));
        const double rdet = 1.0 / det;
        const double dr = hex8_dot(res, nr), ds = hex8_dot(res, ns), dt = hex8_dot(res, nt);
        const double m = max3abs(dr, ds, dt) * fabs(rdet);
        r = fma(dr, rdet, r);
        s = fma(ds, rdet, s);
        t = fma(dt, rdet, t);
        if (!(m >= 1e-9)) break;
        if (!(max3abs(r, s, t) <= 1e3)) break;
        for (int a = 0; a < 3; ++a) hex8_fast_axis(c[a], q0[a], r, s, t, &ar[a], &as[a], &at[a], &res[a]);
    }
    xi[0] = r;
    xi[1] = s;
    xi[2] = t;
}

/* cap: the reference's 50 (:264); the GLL section starts its own iteration from a few trips of this one, with
 * polish != 0: the trip that finds the residual converged still applies its update before it returns. */
static int hex8_newton_capped(const double pnt[3], const double vtx[8][3], double xi[3], int *iters, int cap, int polish)
{
    xi[0] = xi[1] = xi[2] = 0;
    const double sx = fabs(vtx[1][0] - vtx[0][0]);
    const double sy = fabs(vtx[1][1] - vtx[0][1]);
    const double sz = fabs(vtx[1][2] - vtx[0][2]);
    const double sxy = sx > sy ? sx : sy;
    const double scale = sz > sxy ? sz : sxy;
    const double tol = 1e-8 * scale;
    double col[8];
    for (int it = 0; it < cap; ++it) {
        double res[3];
        for (int a = 0; a < 3; ++a) {
            for (int n = 0; n < 8; ++n) col[n] = vtx[n][a];
            res[a] = pnt[a] - hex8_map_axis(col, xi[0], xi[1], xi[2]);
        }
        const int done = fabs(res[0]) < tol && fabs(res[1]) < tol && fabs(res[0]) < tol;
        if (done && !polish) {
            if (iters) *iters = it + 1;
            return 1;
        }
        double inv[3][3];
        hex8_inverse_jacobian(xi, vtx, inv);
        double upd[3];
        for (int a = 0; a < 3; ++a) {
            /* row a of the transposed inverse = column a of inv */
            double acc = 0;
            for (int j = 0; j < 3; ++j) acc = acc + inv[j][a] * res[j];
            upd[a] = acc;
        }
        for (int a = 0; a < 3; ++a) xi[a] = xi[a] + upd[a];
        if (done) {
            if (iters) *iters = it + 1;
            return 1;
        }
    }
    if (iters) *iters = cap;
    return 0;
}

/* A5: hull check (src/trilinearinterpolator.c:157-172): converged and every
 * |xi| <= 2. */
int mmo_hex8_check_hull(const double pnt[3], const double vtx[8][3], double xi[3])
{
    if (!mmo_hex8_newton(pnt, vtx, xi, NULL)) return 0;
    for (int a = 0; a < 3; ++a)
        if (fabs(xi[a]) > (1 + 1.0)) return 0;
    return 1;
}

/* A8: the eight trilinear weights as expanded polynomials
 * (src/trilinearinterpolator.c:174-197).  Sign table per node for the terms
 * rst, rs, rt, r, st, s, t (the constant is always +0.125); summed strictly
 * left to right like the reference. */
void mmo_hex8_weights(const double xi[3], double w[8])
{
    static const signed char sg[8][7] = {
        /* rst  rs  rt   r  st   s   t */
        {-1, +1, +1, -1, +1, -1, -1},
        {+1, -1, +1, -1, -1, +1, -1},
        {-1, +1, -1, +1, -1, +1, -1},
        {+1, -1, -1, +1, +1, -1, -1},
        {+1, +1, -1, -1, -1, -1, +1},
        {-1, -1, +1, +1, -1, -1, +1},
        {+1, +1, +1, +1, +1, +1, +1},
        {-1, -1, -1, -1, +1, +1, +1},
    };
    const double r = xi[0], s = xi[1], t = xi[2];
    for (int n = 0; n < 8; ++n) {
        const signed char *g = sg[n];
        /* (+-0.125 * r * s * t) etc.: products are formed left to right and the
         * sign rides on the exact constant 0.125, as in the reference. */
        double acc = (g[0] * 0.125) * r * s * t;
        acc = acc + (g[1] * 0.125) * r * s;
        acc = acc + (g[2] * 0.125) * r * t;
        acc = acc + (g[3] * 0.125) * r;
        acc = acc + (g[4] * 0.125) * s * t;
        acc = acc + (g[5] * 0.125) * s;
        acc = acc + (g[6] * 0.125) * t;
        acc = acc + 0.125;
        w[n] = acc;
    }
}

static void load_vertices(const i64 *conn, const double *nodes, i64 elem,
                          double vtx[8][3])
{
    for (int n = 0; n < 8; ++n) {
        const i64 id = conn[elem * 8 + n];
        for (int a = 0; a < 3; ++a) vtx[n][a] = nodes[id * 3 + a];
    }
}

/* ------------------------------------------------------------------ */
/* A4: point location + weights, src/trilinearinterpolator.c:40-148.   */
/* Candidates are walked in the given (kNN) order.  Accept the first   */
/* candidate whose hull check passes with max|xi| < 1.025.  Otherwise  */
/* remember the candidate with the smallest max|xi| (strict <, first   */
/* one wins) among those passing the hull check; after the LAST        */
/* candidate, if that smallest value is < 1.5 re-run the hull check on */
/* it and use it, else the point fails.  Failed rows are left          */
/* untouched.  status (optional, our addition for tests): j of the     */
/* accepted candidate, k + j for the fallback candidate, -1 failed.    */
/* ------------------------------------------------------------------ */
i64 mmo_locate_hex8(i64 k, i64 npoints, const i64 *nn, const i64 *conn,
                    i64 *enc, const double *nodes, double *w,
                    const double *points, int *status)
{
    i64 nfailed = 0;
    for (i64 i = 0; i < npoints; ++i) {
        const double *pnt = points + 3 * i;
        double vtx[8][3], xi[3], wt[8];
        double smallest = 99999999.9;
        i64 best = -1;
        int best_j = -1;
        int st = -1;
        int found = 0;
        for (i64 j = 0; j < k; ++j) {
            const i64 elem = nn[i * k + j];
            load_vertices(conn, nodes, elem, vtx);
            if (mmo_hex8_check_hull(pnt, vtx, xi)) {
                double worst = 0.0;
                for (int a = 0; a < 3; ++a)
                    if (fabs(xi[a]) > worst) worst = fabs(xi[a]);
                if (worst < (1 + 0.025)) {
                    mmo_hex8_weights(xi, wt);
                    for (int n = 0; n < 8; ++n) {
                        w[i * 8 + n] = wt[n];
                        enc[i * 8 + n] = conn[elem * 8 + n];
                    }
                    st = (int)j;
                    found = 1;
                    break;
                } else if (worst < smallest) {
                    smallest = worst;
                    best = elem;
                    best_j = (int)j;
                }
            }
        }
        if (!found) {
            /* only reached with j == k-1 exhausted (reference :113, :138) */
            int ok = 0;
            if (k > 0 && smallest < 1.5 && best >= 0) {
                load_vertices(conn, nodes, best, vtx);
                if (mmo_hex8_check_hull(pnt, vtx, xi)) {
                    mmo_hex8_weights(xi, wt);
                    for (int n = 0; n < 8; ++n) {
                        w[i * 8 + n] = wt[n];
                        enc[i * 8 + n] = conn[best * 8 + n];
                    }
                    st = (int)k + best_j;
                    ok = 1;
                }
            }
            /* k == 0: the reference loop body never runs, nothing fails */
            if (!ok && k > 0) nfailed += 1;
        }
        if (status) status[i] = st;
    }
    return nfailed;
}

/* ------------------------------------------------------------------ */
/* A9: weighted gather, scripts/cli.py:98-100:                         */
/*   np.sum(field[enc] * w, axis=1)                                    */
/* Each product is rounded first; the row sum then follows NumPy's     */
/* pairwise add-reduce for a contiguous row of P doubles (P <= 128):   */
/* P < 8: sequential from 0; else 8 running partials r[j], folded as   */
/* ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the P%8 tail in order.    */
/* tests/test_oracle_pinned.py checks this against NumPy itself.       */
/* ------------------------------------------------------------------ */
static double numpy_row_sum(const double *a, i64 n)
{
    if (n < 8) {
        double res = 0.;
        for (i64 i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    i64 i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return 0. + res; /* the reduction starts from the identity: a row sum of -0.0 reads +0.0 */
}

/* field: [ncomp][nsrc] (one contiguous array per component, as the reference
 * keeps one array per parameter); ids: [npoints][P]; w: [npoints][P];
 * out: [npoints][ncomp] when out_point_major, else [ncomp][npoints]. */
int mmo_gather(const double *field, i64 nsrc, i64 ncomp, const i64 *ids,
               const double *w, i64 npoints, i64 P, double *out,
               int out_point_major)
{
    if (P > 128) return -1;
    double prod[128];
    for (i64 c = 0; c < ncomp; ++c) {
        const double *f = field + c * nsrc;
        for (i64 i = 0; i < npoints; ++i) {
            for (i64 p = 0; p < P; ++p)
                prod[p] = f[ids[i * P + p]] * w[i * P + p];
            const double v = numpy_row_sum(prod, P);
            if (out_point_major) out[i * ncomp + c] = v;
            else out[c * npoints + i] = v;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* A2: exact k nearest neighbours by brute force (small cases only).   */
/* The reference calls scipy.spatial.cKDTree(...).query(pts, k)        */
/* (scripts/cli.py:66-73); scipy is a third-party dependency, so this  */
/* restates its published contract: the k smallest Euclidean distances */
/* in ascending order, squared distance accumulated axis by axis       */
/* ((dx*dx + dy*dy) + dz*dz, no fused multiply-add).  Equal distances  */
/* are ordered by index here (cKDTree's tie order is unspecified).     */
/* Rows with fewer than k sources are padded with index nsrc and +inf  */
/* like cKDTree.                                                       */
/* ------------------------------------------------------------------ */
void mmo_knn_brute(const double *src, i64 nsrc, const double *pts, i64 npts,
                   i64 ndim, i64 k, i64 *idx, double *d2out)
{
    double *bd = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    i64 *bi = (i64 *)malloc(sizeof(i64) * (size_t)(k > 0 ? k : 1));
    for (i64 i = 0; i < npts; ++i) {
        i64 have = 0;
        for (i64 e = 0; e < nsrc; ++e) {
            double d2 = 0.;
            for (i64 a = 0; a < ndim; ++a) {
                const double d = src[e * ndim + a] - pts[i * ndim + a];
                d2 += d * d;
            }
            if (have == k && !(d2 < bd[k - 1])) continue; /* ties keep lower index */
            i64 pos = have < k ? have : k - 1;
            while (pos > 0 && d2 < bd[pos - 1]) {
                bd[pos] = bd[pos - 1];
                bi[pos] = bi[pos - 1];
                --pos;
            }
            bd[pos] = d2;
            bi[pos] = e;
            if (have < k) ++have;
        }
        for (i64 j = 0; j < k; ++j) {
            idx[i * k + j] = j < have ? bi[j] : nsrc;
            if (d2out) d2out[i * k + j] = j < have ? bd[j] : INFINITY;
        }
    }
    free(bd);
    free(bi);
}

/* ================================================================== */
/* A10: GLL elements (order 1, 2, 4; 2-D and 3-D).  PARITY UNPINNED:   */
/* the reference obtains the inverse coordinate transform and the      */
/* interpolation coefficients from the proprietary salvus.fem package  */
/* (components/interpolator.py:22-57, :1337-1347, :1370-1386), which   */
/* is not available; no reference test pins its results.  The CONTROL  */
/* FLOW below follows the reference (interpolator.py:1181-1233); the   */
/* NUMERICS are this project's own definition, shared verbatim with    */
/* the HIP kernel (multimesh_amd/csrc/mm_locate_gll.hip):              */
/*  - tensor-product Lagrange basis on the GLL nodes of [-1,1]         */
/*    (order 1: -1,1; order 2: -1,0,1; order 4: -1,-sqrt(3/7),0,       */
/*    sqrt(3/7),1), node index p = i + (n+1) j + (n+1)^2 k with xi_1   */
/*    fastest;                                                         */
/*  - Newton on x(xi) = sum_p L_p(xi) X_p from xi = 0 -- in 3-D at     */
/*    order >= 2 from the solution of the eight corners' trilinear map */
/*    (<= 8 trips of the hex8 iteration of section A6, used when       */
/*    finite and within |xi| <= 3) --, Jacobian by the                 */
/*    analytic basis derivatives, cofactor solve; converged when the   */
/*    largest component of the update is < 1e-10; at most 25 updates;  */
/*    NaN when the Jacobian is singular, an iterate leaves [-10,10]    */
/*    or the iteration does not converge.                              */
/* ================================================================== */
static void gll_nodes(int order, double *g)
{
    if (order == 1) { g[0] = -1.0; g[1] = 1.0; }
    else if (order == 2) { g[0] = -1.0; g[1] = 0.0; g[2] = 1.0; }
    else { const double a = 0x1.4f2ec413cb52ap-1; /* sqrt(3/7) */ g[0] = -1.0; g[1] = -a; g[2] = 0.0; g[3] = a; g[4] = 1.0; }
}

/* 1-D Lagrange values l[i] and derivatives dl[i] at x.  Product formulas with a fixed loop order
 * and PRECOMPUTED reciprocals of the node differences (no divisions in the hot loop); the HIP
 * kernel does exactly the same operations:
 *   e[i][m] = (x - g[m]) * (1 / (g[i] - g[m]))
 *   l[i]    = prod_{m != i} e[i][m]                      (m ascending, starting from 1.0)
 *   dl[i]   = sum_{m != i} (1/(g[i]-g[m])) * prod_{q != i,m} e[i][q] */
static void lagrange_1d(int order, const double *g, double x, double *l, double *dl)
{
    const int n = order + 1;
    double inv[5][5], e[5][5];
    for (int i = 0; i < n; ++i)
        for (int m = 0; m < n; ++m)
            if (m != i) {
                inv[i][m] = 1.0 / (g[i] - g[m]);
                e[i][m] = (x - g[m]) * inv[i][m];
            }
    for (int i = 0; i < n; ++i) {
        double v = 1.0;
        for (int m = 0; m < n; ++m)
            if (m != i) v = v * e[i][m];
        l[i] = v;
        double d = 0.0;
        for (int m = 0; m < n; ++m) {
            if (m == i) continue;
            double t = inv[i][m];
            for (int q = 0; q < n; ++q)
                if (q != i && q != m) t = t * e[i][q];
            d = d + t;
        }
        dl[i] = d;
    }
}

/* interpolation coefficients at xi (reference get_coefficients, interpolator.py:1337-1347) */
void mmo_gll_coefficients(int order, int dim, const double *xi, double *coeffs)
{
    double g[5], l[3][5], dl[3][5];
    const int n = order + 1;
    gll_nodes(order, g);
    for (int d = 0; d < dim; ++d) lagrange_1d(order, g, xi[d], l[d], dl[d]);
    if (dim == 3) {
        for (int k = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) coeffs[i + n * (j + n * k)] = (l[0][i] * l[1][j]) * l[2][k];
    } else {
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) coeffs[i + n * j] = l[0][i] * l[1][j];
    }
}

/* An INDEPENDENT statement of the same inverse transform, for bounding the production arithmetic below (it is
 * not what the HIP kernels are compared with bit for bit): the map and its Jacobian as plain sums over the P
 * nodes of L_p(xi) X_p with L_p = (l0[i] l1[j]) l2[k] -- no sum factorisation, no fused multiply-add --, a
 * Gaussian solve with partial pivoting instead of the cofactor inverse, converged at |update| < 1e-13, at most
 * 50 updates.  mmo_set_gll_strict(1) routes mmo_gll_inverse_transform (and with it both acceptance loops) through
 * it: tests/test_gll_oracle.py asserts that the accepted elements are the same and that xi and the coefficients
 * agree to 1e-9, so that a change of the production arithmetic (the stop test went from 1e-12 to 1e-10 and the
 * sums to fma in round 2) is bounded by something other than itself. */
static int g_gll_strict = 0;
void mmo_set_gll_strict(int on) { g_gll_strict = on ? 1 : 0; }

static void gll_inverse_transform_strict(int order, int dim, const double *pnt, const double *ctrl, double *xi)
{
    double g[5], l[3][5], dl[3][5];
    const int n = order + 1;
    gll_nodes(order, g);
    for (int d = 0; d < dim; ++d) xi[d] = 0.0;
    for (int d = dim; d < 3; ++d) { l[d][0] = 1.0; dl[d][0] = 0.0; }
    const int nk = dim == 3 ? n : 1;
    for (int it = 0; it < 50; ++it) {
        for (int d = 0; d < dim; ++d) lagrange_1d(order, g, xi[d], l[d], dl[d]);
        double A[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   /* [J | r] */
        for (int k = 0; k < nk; ++k)
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    const double *X = ctrl + dim * (i + n * (j + n * k));
                    const double lk = dim == 3 ? l[2][k] : 1.0, dk = dim == 3 ? dl[2][k] : 0.0;
                    const double w = l[0][i] * l[1][j] * lk;
                    const double w0 = dl[0][i] * l[1][j] * lk, w1 = l[0][i] * dl[1][j] * lk, w2 = l[0][i] * l[1][j] * dk;
                    for (int a = 0; a < dim; ++a) {
                        A[a][3] += w * X[a];
                        A[a][0] += w0 * X[a];
                        A[a][1] += w1 * X[a];
                        if (dim == 3) A[a][2] += w2 * X[a];
                    }
                }
        for (int a = 0; a < dim; ++a) A[a][3] -= pnt[a];
        if (dim == 2) { A[0][2] = A[0][3]; A[1][2] = A[1][3]; }
        /* Gaussian elimination with partial pivoting on the dim x (dim + 1) system J dxi = r */
        int singular = 0;
        for (int c = 0; c < dim && !singular; ++c) {
            int piv = c;
            for (int r = c + 1; r < dim; ++r)
                if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
            if (!(fabs(A[piv][c]) > 0.0)) { singular = 1; break; }
            for (int q = 0; q <= dim; ++q) { const double t = A[c][q]; A[c][q] = A[piv][q]; A[piv][q] = t; }
            for (int r = c + 1; r < dim; ++r) {
                const double f = A[r][c] / A[c][c];
                for (int q = c; q <= dim; ++q) A[r][q] -= f * A[c][q];
            }
        }
        if (singular) break;
        double dxi[3] = {0, 0, 0};
        for (int c = dim - 1; c >= 0; --c) {
            double v = A[c][dim];
            for (int q = c + 1; q < dim; ++q) v -= A[c][q] * dxi[q];
            dxi[c] = v / A[c][c];
        }
        double step = 0.0;
        int bad = 0;
        for (int a = 0; a < dim; ++a) {
            xi[a] = xi[a] - dxi[a];
            if (fabs(dxi[a]) > step) step = fabs(dxi[a]);
            if (!(fabs(xi[a]) <= 10.0)) bad = 1;
        }
        if (bad) break;
        if (step < 1e-13) return;
    }
    for (int d = 0; d < dim; ++d) xi[d] = NAN;
}

/* inverse coordinate transform (reference inverse_transform, interpolator.py:1370-1386): ctrl is
 * the element's P control nodes [P][dim]; xi receives the reference coordinates or NaNs. */
void mmo_gll_inverse_transform(int order, int dim, const double *pnt, const double *ctrl, double *xi)
{
    if (g_gll_strict) {
        gll_inverse_transform_strict(order, dim, pnt, ctrl, xi);
        return;
    }
    double g[5], l[3][5], dl[3][5];
    const int n = order + 1;
    gll_nodes(order, g);
    for (int d = 0; d < dim; ++d) xi[d] = 0.0;
    if (dim == 3 && order >= 2) {
        /* Start from the solution of the eight CORNERS' trilinear map: at most 8 trips of mmo_hex8_start above (rounds 2-3:
         * of the reference-order hex8 iteration, the converged trip's update applied as well) (corner c of trilinearinterpolator.c:8-10 is the control node at the matching end of every axis).  A start
         * that is not finite or lies beyond 3 is not used.  Part of this path's definition, like the fma order
         * below: the HIP kernel does the same. */
        double vtx[8][3], q[3];
        for (int c = 0; c < 8; ++c) {
            const int node = (kR[c] > 0 ? n - 1 : 0) + n * ((kS[c] > 0 ? n - 1 : 0) + n * (kT[c] > 0 ? n - 1 : 0));
            for (int a = 0; a < 3; ++a) vtx[c][a] = ctrl[3 * node + a];
        }
        mmo_hex8_start(pnt, (const double(*)[3])vtx, q, 8);
        if (fabs(q[0]) <= 3.0 && fabs(q[1]) <= 3.0 && fabs(q[2]) <= 3.0)
            for (int d = 0; d < 3; ++d) xi[d] = q[d];
    }
    for (int it = 0; it < 25; ++it) {
        for (int d = 0; d < dim; ++d) lagrange_1d(order, g, xi[d], l[d], dl[d]);
        double x[3] = {0, 0, 0}, J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        /* Tensor-product (sum-factorised) evaluation of the map and its Jacobian; the order of the
         * partial sums below IS the definition of this path's arithmetic (the HIP kernel follows it
         * operation for operation): innermost over i with l0 / dl0, then over j, then over k, every partial
         * sum accumulated with one fused multiply-add (C99 fma(): correctly rounded, so identical to
         * the GPU's v_fma_f64). */
        if (dim == 3) {
            for (int k = 0; k < n; ++k) {
                double b00[3] = {0, 0, 0}, b01[3] = {0, 0, 0}, b10[3] = {0, 0, 0};
                for (int j = 0; j < n; ++j) {
                    double a0[3] = {0, 0, 0}, a1[3] = {0, 0, 0};
                    for (int i = 0; i < n; ++i) {
                        const double *X = ctrl + 3 * (i + n * (j + n * k));
                        for (int a = 0; a < 3; ++a) {
                            a0[a] = fma(l[0][i], X[a], a0[a]);
                            a1[a] = fma(dl[0][i], X[a], a1[a]);
                        }
                    }
                    for (int a = 0; a < 3; ++a) {
                        b00[a] = fma(l[1][j], a0[a], b00[a]);
                        b01[a] = fma(dl[1][j], a0[a], b01[a]);
                        b10[a] = fma(l[1][j], a1[a], b10[a]);
                    }
                }
                for (int a = 0; a < 3; ++a) {
                    x[a] = fma(l[2][k], b00[a], x[a]);
                    J[a][0] = fma(l[2][k], b10[a], J[a][0]);
                    J[a][1] = fma(l[2][k], b01[a], J[a][1]);
                    J[a][2] = fma(dl[2][k], b00[a], J[a][2]);
                }
            }
        } else {
            for (int j = 0; j < n; ++j) {
                double a0[2] = {0, 0}, a1[2] = {0, 0};
                for (int i = 0; i < n; ++i) {
                    const double *X = ctrl + 2 * (i + n * j);
                    for (int a = 0; a < 2; ++a) {
                        a0[a] = fma(l[0][i], X[a], a0[a]);
                        a1[a] = fma(dl[0][i], X[a], a1[a]);
                    }
                }
                for (int a = 0; a < 2; ++a) {
                    x[a] = fma(l[1][j], a0[a], x[a]);
                    J[a][0] = fma(l[1][j], a1[a], J[a][0]);
                    J[a][1] = fma(dl[1][j], a0[a], J[a][1]);
                }
            }
        }
        double r[3] = {0, 0, 0}, dxi[3] = {0, 0, 0};
        for (int a = 0; a < dim; ++a) r[a] = x[a] - pnt[a];
        if (dim == 3) {
            const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
            const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
            const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            const double det = (J[0][0] * c00 + J[0][1] * c01) + J[0][2] * c02;
            const double rdet = 1.0 / det;
            /* inverse = adjugate / det; dxi = inverse * r */
            const double i00 = c00 * rdet, i01 = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rdet,
                         i02 = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rdet;
            const double i10 = c01 * rdet, i11 = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rdet,
                         i12 = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rdet;
            const double i20 = c02 * rdet, i21 = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rdet,
                         i22 = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rdet;
            dxi[0] = (i00 * r[0] + i01 * r[1]) + i02 * r[2];
            dxi[1] = (i10 * r[0] + i11 * r[1]) + i12 * r[2];
            dxi[2] = (i20 * r[0] + i21 * r[1]) + i22 * r[2];
        } else {
            const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
            const double rdet = 1.0 / det;
            dxi[0] = (J[1][1] * r[0] - J[0][1] * r[1]) * rdet;
            dxi[1] = (J[0][0] * r[1] - J[1][0] * r[0]) * rdet;
        }
        double step = 0.0;
        int bad = 0;
        for (int a = 0; a < dim; ++a) {
            xi[a] = xi[a] - dxi[a];
            if (fabs(dxi[a]) > step) step = fabs(dxi[a]);
            if (!(fabs(xi[a]) <= 10.0)) bad = 1; /* also catches NaN */
        }
        if (bad) break;
        if (step < 1e-10) return; /* (the update just applied: what is left is its square) */
    }
    for (int d = 0; d < dim; ++d) xi[d] = NAN;
}

/* Element search + coefficients, control flow of reference get_element_weights.check_inside
 * (interpolator.py:1181-1233): walk the candidates in order; skip NaN results; remember the
 * candidate with the smallest max|xi| (the first finite one always replaces the 10e9 start value);
 * accept the first with all |xi| < tolerance; otherwise, with snap_to_nearest, clip the remembered
 * xi to +-1.02 and use it, else element -1 and zero coefficients.
 * gll_points [nelem][P][dim]; nn [npoints][k]; elem [npoints]; coeffs [npoints][P].
 * Returns the number of points that got element -1. */
i64 mmo_locate_gll(int order, int dim, i64 k, i64 npoints, const i64 *nn, const double *gll_points, i64 nelem,
                   const double *points, double tolerance, int snap_to_nearest, i64 *elem, double *coeffs)
{
    int P = 1;
    for (int d = 0; d < dim; ++d) P *= order + 1;
    i64 missing = 0;
    for (i64 i = 0; i < npoints; ++i) {
        const double *pnt = points + i * dim;
        double best_xi[3] = {10e9, 10e9, 10e9};
        double best_val = 10e9;
        i64 best_elem = 0;
        int found = 0;
        double xi[3];
        for (i64 j = 0; j < k && !found; ++j) {
            const i64 e = nn[i * k + j];
            if (e < 0 || e >= nelem) continue;
            mmo_gll_inverse_transform(order, dim, pnt, gll_points + (size_t)e * P * dim, xi);
            int isnan_any = 0;
            double worst = 0.0;
            for (int d = 0; d < dim; ++d) {
                if (xi[d] != xi[d]) isnan_any = 1;
                if (fabs(xi[d]) > worst) worst = fabs(xi[d]);
            }
            if (isnan_any) continue;
            if (worst < best_val) {
                best_val = worst;
                best_elem = e;
                for (int d = 0; d < dim; ++d) best_xi[d] = xi[d];
            }
            int inside = 1;
            for (int d = 0; d < dim; ++d)
                if (!(fabs(xi[d]) < tolerance)) inside = 0;
            if (inside) {
                elem[i] = e;
                mmo_gll_coefficients(order, dim, xi, coeffs + i * P);
                found = 1;
            }
        }
        if (found) continue;
        if (snap_to_nearest) {
            for (int d = 0; d < dim; ++d) {
                double v = best_xi[d];
                if (v < -1.02) v = -1.02;
                if (v > 1.02) v = 1.02;
                best_xi[d] = v;
            }
            elem[i] = best_elem;
            mmo_gll_coefficients(order, dim, best_xi, coeffs + i * P);
        } else {
            elem[i] = -1;
            for (int p = 0; p < P; ++p) coeffs[i * P + p] = 0.0;
            missing += 1;
        }
    }
    return missing;
}

/* A10, variant 1: _check_if_inside_element + boundary_box_check (reference interpolator.py:1350-1367,
 * :1409-1473; used by gll_2_exodus :274, gll_2_gll_layered_multi :543 and the helpers :1523, :1572).
 * Per candidate, in order: axis-aligned bounding box of the control nodes (inclusive); inside ->
 * inverse transform, skip NaN, accept when every |xi| <= 1.04.  Nothing accepted: take the FIRST
 * candidate whose box contains the point (their "distance" is 0, np.where(dist == min)[0][0]), or,
 * if no box does, the candidate whose control-node mean is nearest (first minimum); transform again;
 * NaN (the reference raises unless ignore_hard_elements) or any |xi| >= 1.04 -> the constant
 * xi = (0.645, -0.5, 0.22) (:1468-1471; its first `dim` components).  Output: element id and the P
 * Lagrange coefficients of xi.  PARITY UNPINNED like the rest of the GLL section; our definitions:
 * the centre is the sequential sum of the P nodes from 0.0 divided by P, distances are compared
 * squared.  Returns the number of points whose final transform was NaN (the reference's ValueError
 * cases).  Points without any valid candidate get element -1 and zero coefficients (and count). */
void mmo_gll_element_boxes(int dim, i64 nelem, i64 P, const double *gll_points, double *boxes)
{
    for (i64 e = 0; e < nelem; ++e) {
        const double *X = gll_points + (size_t)e * P * dim;
        double *b = boxes + (size_t)e * 3 * dim; /* min[dim], max[dim], centre[dim] */
        for (int d = 0; d < dim; ++d) {
            double mn = X[d], mx = X[d], sum = 0.0;
            for (i64 p = 0; p < P; ++p) {
                const double v = X[p * dim + d];
                if (v < mn) mn = v;
                if (v > mx) mx = v;
                sum = sum + v;
            }
            b[d] = mn;
            b[dim + d] = mx;
            b[2 * dim + d] = sum / (double)P;
        }
    }
}

i64 mmo_locate_gll_v1(int order, int dim, i64 k, i64 npoints, const i64 *nn, const double *gll_points, i64 nelem,
                      const double *points, i64 *elem, double *coeffs)
{
    static const double hard_xi[3] = {0.645, -0.5, 0.22};
    int P = 1;
    for (int d = 0; d < dim; ++d) P *= order + 1;
    double *boxes = (double *)malloc(sizeof(double) * (size_t)(nelem > 0 ? nelem : 1) * 3 * dim);
    mmo_gll_element_boxes(dim, nelem, P, gll_points, boxes);
    i64 hard = 0;
    for (i64 i = 0; i < npoints; ++i) {
        const double *pnt = points + i * dim;
        i64 first_inside = -1, nearest = -1;
        double nearest_d2 = INFINITY;
        int found = 0;
        double xi[3];
        for (i64 j = 0; j < k && !found; ++j) {
            const i64 e = nn[i * k + j];
            if (e < 0 || e >= nelem) continue;
            const double *b = boxes + (size_t)e * 3 * dim;
            int inside = 1;
            for (int d = 0; d < dim; ++d)
                if (!(pnt[d] >= b[d] && pnt[d] <= b[dim + d])) inside = 0;
            if (inside) {
                if (first_inside < 0) first_inside = j;
                mmo_gll_inverse_transform(order, dim, pnt, gll_points + (size_t)e * P * dim, xi);
                int ok = 1;
                for (int d = 0; d < dim; ++d)
                    if (!(fabs(xi[d]) <= 1.04)) ok = 0; /* NaN fails the comparison too */
                if (ok) {
                    elem[i] = e;
                    mmo_gll_coefficients(order, dim, xi, coeffs + i * P);
                    found = 1;
                }
            } else {
                double d2 = 0.0;
                for (int d = 0; d < dim; ++d) {
                    const double t = pnt[d] - b[2 * dim + d];
                    d2 = d2 + t * t;
                }
                if (d2 < nearest_d2) {
                    nearest_d2 = d2;
                    nearest = j;
                }
            }
        }
        if (found) continue;
        const i64 ind = first_inside >= 0 ? first_inside : nearest;
        if (ind < 0) {
            elem[i] = -1;
            for (int p = 0; p < P; ++p) coeffs[i * P + p] = 0.0;
            hard += 1;
            continue;
        }
        const i64 e = nn[i * k + ind];
        mmo_gll_inverse_transform(order, dim, pnt, gll_points + (size_t)e * P * dim, xi);
        int isnan_any = 0, far = 0;
        for (int d = 0; d < dim; ++d) {
            if (xi[d] != xi[d]) isnan_any = 1;
            if (fabs(xi[d]) >= 1.04) far = 1;
        }
        if (isnan_any) hard += 1;
        if (isnan_any || far)
            for (int d = 0; d < dim; ++d) xi[d] = hard_xi[d];
        elem[i] = e;
        mmo_gll_coefficients(order, dim, xi, coeffs + i * P);
    }
    free(boxes);
    return hard;
}

/* Element-nodal gather: np.sum(coeffs * field[elem_indices], axis=1) (reference
 * interpolator.py:976); field [ncomp][nelem][P]; element -1 contributes zeros (signed like NumPy's). */
int mmo_gather_elem(const double *field, i64 nelem, i64 ncomp, const i64 *elem, const double *coeffs, i64 npoints,
                    i64 P, double *out, int out_point_major)
{
    if (P > 128) return -1;
    double prod[128];
    for (i64 c = 0; c < ncomp; ++c) {
        const double *f = field + c * nelem * P;
        for (i64 i = 0; i < npoints; ++i) {
            /* NumPy index semantics: -1 (not found, zero coefficients) reads the last element */
            const i64 e = elem[i] >= 0 && elem[i] < nelem ? elem[i] : (elem[i] < 0 && elem[i] >= -nelem ? elem[i] + nelem : 0);
            for (i64 p = 0; p < P; ++p) prod[p] = coeffs[i * P + p] * f[e * P + p];
            const double v = numpy_row_sum(prod, P);
            if (out_point_major) out[i * ncomp + c] = v;
            else out[c * npoints + i] = v;
        }
    }
    return 0;
}
