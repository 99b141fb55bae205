// A4..A8 -- hex8 point location by Newton inversion + trilinear weights.
// Replaces reference multi_mesh/src/trilinearinterpolator.c:40-148 (triLinearInterpolator) and
// its helpers :150-375.
//
// One lane per target point.  The lane walks its candidate elements in the given (kNN) order;
// for each it gathers the 8 corner ids and coordinates, runs the fp64 Newton iteration entirely
// in registers and applies the reference's acceptance rules:
//   * hull check: converged (<= 50 iterations, tolerance 1e-8 * max|v1-v0|, residual test on
//     x, y and x again -- z is never tested, trilinearinterpolator.c:290-291) and all |xi| <= 2;
//   * accept the first candidate with max|xi| < 1.025 (:93);
//   * otherwise remember the candidate with the smallest max|xi| (:105-110) and, after the last
//     candidate, re-run it and use it if that value is < 1.5 (:113-132); else the point fails
//     (:133-143) and its output rows are left untouched.
// Every floating-point expression keeps the reference's association order and the file is built
// with -ffp-contract=off, so node ids AND weights are bit-identical to the reference.
//
// Scheduling.  Walking the candidates in lockstep makes every wave pay for its unluckiest lane
// in every round (only ~67 % of mesh-node targets are accepted at the nearest centroid, 1.6 solves
// per target on average, and a diverging Newton runs to the 50-iteration cap).  So a solve is the
// unit of scheduling: persistent waves perform ONE Newton solve per lane and round, keep their
// unresolved targets in a private LDS queue and retry them, densely packed, on their next
// candidate (locate_pass_kernel below).  Before a solve a lane skips candidates whose corner
// bounding box (x and y only, widened by 5 %) cannot contain the point: such a candidate can
// never be ACCEPTED -- acceptance needs max|xi| < 1.025 and converged x/y residuals, and the
// trilinear image of [-1.025,1.025]^3 stays within 3*0.025*(1.025^2/2) = 3.94 % of the corner
// extent outside the corner bounding box (z is excluded because the reference never tests the z
// residual).  Skipping is therefore invisible whenever some candidate is accepted.  A solve that
// needs more than kPassIters iterations is parked and resumed by a second launch with the
// reference's cap of 50.  Only a target that runs out of candidates without an acceptance (so the
// reference's "least outside" fallback or a failure is due) is handed to the reference-order
// kernel, which redoes it from scratch.
//
// HBM-bound by the roofline accounting (568 B/target when the first candidate is accepted:
// 24 point + 8k candidates + 64 connectivity row + 192 corner coordinates + 128 out), though
// the lane spends most of its time in the dependent Newton chain; corner gathers hit L2.
#include <cstring>

#include "mm_common.h"
#include "mm_newton_hex8.h"

namespace {

// map_axis, newton_hex8 and the corner signs MM_R / MM_S / MM_T: mm_newton_hex8.h (shared with the host-side test of
// its arithmetic, tests/test_newton_host.py)

// Eight weights as expanded polynomials (trilinearinterpolator.c:174-197); the sign rides on
// the exact constant 0.125 and the terms are added strictly left to right.
__device__ __forceinline__ void weights_hex8(const double (&xi)[3], double (&w)[8])
{
    const double r = xi[0], s = xi[1], t = xi[2];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        // signs: rst = R*S*T, rs = -(R*S*T)*T ... derived from the node's corner signs
        const double R = MM_R(n), S = MM_S(n), T = MM_T(n);
        const double c_rst = 0.125 * (R * S * T);
        const double c_rs = 0.125 * (R * S);
        const double c_rt = 0.125 * (R * T);
        const double c_r = 0.125 * R;
        const double c_st = 0.125 * (S * T);
        const double c_s = 0.125 * S;
        const double c_t = 0.125 * T;
        double acc = c_rst * r * s * t;
        acc = acc + c_rs * r * s;
        acc = acc + c_rt * r * t;
        acc = acc + c_r * r;
        acc = acc + c_st * s * t;
        acc = acc + c_s * s;
        acc = acc + c_t * t;
        acc = acc + 0.125;
        w[n] = acc;
    }
}

// ID: i64 as the arrays hold them, or int where the caller has checked that node ids fit (half the registers: the
// pass kernel keeps the ids across the Newton solve)
template <typename ID>
struct CornersT {
    ID id[8];
    double x[8], y[8], z[8];
};
typedef CornersT<i64> Corners;

// the element's eight node ids in the locator's corner order ...
template <bool EXODUS, typename ID>
__device__ __forceinline__ void load_ids(const i64 *__restrict__ conn, i64 elem, ID (&id)[8])
{
    const longlong2 *row = reinterpret_cast<const longlong2 *>(conn + elem * 8);
    const longlong2 a = row[0], b = row[1], cc = row[2], d = row[3];
    id[0] = (ID)a.x;
    id[1] = (ID)(EXODUS ? b.y : a.y);  // reference scripts/cli.py:79-81: columns 1 and 3 swap
    id[2] = (ID)b.x;
    id[3] = (ID)(EXODUS ? a.y : b.y);
    id[4] = (ID)cc.x;
    id[5] = (ID)cc.y;
    id[6] = (ID)d.x;
    id[7] = (ID)d.y;
}

// ... and their coordinates
template <typename ID>
__device__ __forceinline__ void load_xyz(const double *__restrict__ nodes, CornersT<ID> &c)
{
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const double *p = nodes + (i64)c.id[n] * 3;
        c.x[n] = p[0];
        c.y[n] = p[1];
        c.z[n] = p[2];
    }
}

template <bool EXODUS, typename ID>
__device__ __forceinline__ void load_corners(const i64 *__restrict__ conn,
                                             const double *__restrict__ nodes, i64 elem, CornersT<ID> &c)
{
    load_ids<EXODUS>(conn, elem, c.id);
    load_xyz(nodes, c);
}

__device__ __forceinline__ double max_abs3(const double (&xi)[3])
{
    double worst = 0.0;
    if (fabs(xi[0]) > worst) worst = fabs(xi[0]);
    if (fabs(xi[1]) > worst) worst = fabs(xi[1]);
    if (fabs(xi[2]) > worst) worst = fabs(xi[2]);
    return worst;
}

__device__ __forceinline__ bool in_hull(const double (&xi)[3])
{
    return !(fabs(xi[0]) > (1 + 1.0)) && !(fabs(xi[1]) > (1 + 1.0)) && !(fabs(xi[2]) > (1 + 1.0));
}

// What happens to a located point.  enc/w non-null: the operator rows are stored (the reference's
// output).  out non-null: the weighted sum of the element's nodal values is formed right here (A9
// fused into A4: the fused pipeline then neither writes the 128-byte operator rows nor reads them
// back in a gather launch).  Same arithmetic as mm_gather for P = 8: every product rounded on its
// own, summed as ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) -- NumPy's pairwise order for an 8-term row;
// out-of-range node ids read node 0 like there.
struct Emit {
    i64 *enc;
    double *w;
    const double *fields;   // [ncomp][nnodes]
    i64 nnodes;
    int ncomp;
    double *out;            // [npoints][ncomp]
};

// FMA (MM_FP_TOL only): the sum as one chain of fused multiply-adds -- within a few ulp of NumPy's order.
template <bool FMA = false, typename ID>
__device__ __forceinline__ void emit_row(const Emit &em, i64 i, const ID (&id)[8], const double (&wt)[8])
{
    if (em.enc) {
        longlong2 *e2 = reinterpret_cast<longlong2 *>(em.enc + i * 8);
        double2 *w2 = reinterpret_cast<double2 *>(em.w + i * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e2[q] = make_longlong2((i64)id[2 * q], (i64)id[2 * q + 1]);
            w2[q] = make_double2(wt[2 * q], wt[2 * q + 1]);
        }
    }
    if (em.out) {
        i64 sid[8];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            sid[n] = (unsigned long long)(i64)id[n] < (unsigned long long)em.nnodes ? (i64)id[n] : 0;
        for (int c = 0; c < em.ncomp; ++c) {
            const double *f = em.fields + (i64)c * em.nnodes;
            if (FMA) {
                double v[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) v[n] = f[sid[n]];
                double acc = 0.0;
#pragma unroll
                for (int n = 0; n < 8; ++n) acc = __builtin_fma(v[n], wt[n], acc);
                em.out[i * em.ncomp + c] = acc;
                continue;
            }
            double p[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) p[n] = f[sid[n]] * wt[n];
            // 0.0 + ...: NumPy starts the reduction from the identity (a row sum of -0.0 reads +0.0)
            em.out[i * em.ncomp + c] = 0.0 + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        }
    }
}

template <bool EXODUS, typename IDX>
__global__ __launch_bounds__(256) void locate_hex8_kernel(i64 k, i64 npoints,
                                                          const IDX *__restrict__ nn,
                                                          const i64 *__restrict__ conn, i64 nelem,
                                                          Emit em,
                                                          const double *__restrict__ nodes,
                                                          const double *__restrict__ pts,
                                                          unsigned long long *__restrict__ nfailed,
                                                          const int *__restrict__ list,
                                                          const int *__restrict__ list_count, int zero_failed,
                                                          const int *__restrict__ abort6 = nullptr, int list_min = -1)
{
    if (mm_aborted(abort6)) return;   // (a guessed grid that is not this call's: the host runs the call again)
    // list != null: only the queued targets (left over by the fast passes), grid-stride
    const i64 total = list ? (i64)*list_count : npoints;
    if (list && total <= (i64)list_min) return;   // (shorter lists: locate_hex8_group_kernel, one candidate per lane)
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 q0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    // every lane of a wave runs the same number of trips so that the ballot below is wave-wide
    const i64 trips = (total + stride - 1) / stride;
    for (i64 trip = 0; trip < trips; ++trip) {
    const i64 q = q0 + trip * stride;
    const i64 i = q < total ? (list ? (i64)list[q] : q) : npoints;
    bool failed = false;
    if (i < npoints && k > 0) {
        const double px = pts[i * 3 + 0], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
        double smallest = 99999999.9;
        i64 best = -1;
        bool found = false;
        Corners c;
        double xi[3], wt[8];
        for (i64 j = 0; j < k; ++j) {
            const i64 elem = (i64)nn[i * k + j];
            if (nelem > 0 && (unsigned long long)elem >= (unsigned long long)nelem) continue;
            load_corners<EXODUS>(conn, nodes, elem, c);
            if (newton_hex8(px, py, pz, c.x, c.y, c.z, xi) && in_hull(xi)) {
                const double worst = max_abs3(xi);
                if (worst < (1 + 0.025)) {
                    weights_hex8(xi, wt);
                    emit_row(em, i, c.id, wt);
                    found = true;
                    break;
                } else if (worst < smallest) {
                    smallest = worst;
                    best = elem;
                }
            }
        }
        if (!found) {
            bool ok = false;
            if (smallest < 1.5 && best >= 0) {
                load_corners<EXODUS>(conn, nodes, best, c);
                if (newton_hex8(px, py, pz, c.x, c.y, c.z, xi) && in_hull(xi)) {
                    weights_hex8(xi, wt);
                    emit_row(em, i, c.id, wt);
                    ok = true;
                }
            }
            failed = !ok;
            if (failed && zero_failed) {
                // fused pipeline: its outputs are not pre-zeroed; a failed row reads as zero ids and
                // zero weights (scripts/cli.py:77-78), and its value is the gather of such a row
                const i64 zid[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const double zw[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                emit_row(em, i, zid, zw);
            }
        }
    }
    const unsigned long long mask = __ballot(failed);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(nfailed, (unsigned long long)__popcll(mask));
    }
}

// The same verdicts with ONE CANDIDATE PER LANE: G lanes share a target, lane j solves candidate j (every solve starts
// from xi = 0 and is deterministic, so its result does not depend on who runs it), then the group takes what the loop
// above would have taken -- the first candidate, in list order, accepted below 1.025; else the first one of smallest
// max |xi| among those that converged inside the hull, if below 1.5 (the loop solves that one again: same iterates, same
// weights).  The targets that reach this kernel have found no acceptance in the passes: the loop walks all k candidates
// for each of them, up to the reference's 50 trips apiece, on one lane -- on a graded 10M mesh the slowest lanes made the
// launch last 0.5 - 1 ms for a few tens of thousands of targets (u^2.2: 2.9 ms); here a target's critical path is one solve.
// Lists longer than list_max keep the loop (as many lanes as targets fill the chip there, at full lane use).
template <bool EXODUS, typename IDX, int G>
__global__ __launch_bounds__(256) void locate_hex8_group_kernel(i64 k, i64 npoints, const IDX *__restrict__ nn,
                                                                const i64 *__restrict__ conn, i64 nelem, Emit em,
                                                                const double *__restrict__ nodes,
                                                                const double *__restrict__ pts,
                                                                unsigned long long *__restrict__ nfailed,
                                                                const int *__restrict__ list,
                                                                const int *__restrict__ list_count, int zero_failed,
                                                                int list_max, const int *__restrict__ abort6 = nullptr)
{
    static_assert(G == 32 || G == 64, "a group is a wave or half of one");
    if (mm_aborted(abort6)) return;
    const i64 total = (i64)*list_count;
    if (total > (i64)list_max) return;
    const int lane = threadIdx.x & 63, sub = threadIdx.x & (G - 1);
    const i64 per_block = 256 / G;
    const i64 stride = (i64)gridDim.x * per_block;
    const i64 trips = (total + stride - 1) / stride;
    for (i64 trip = 0; trip < trips; ++trip) {
        const i64 q = (i64)blockIdx.x * per_block + threadIdx.x / G + trip * stride;
        const i64 i = q < total ? (i64)list[q] : npoints;
        const bool live = i < npoints && k > 0;
        bool ok = false;
        double worst = INFINITY;
        Corners c;
        double xi[3] = {0.0, 0.0, 0.0};
        if (live && sub < k) {
            const i64 elem = (i64)nn[i * k + sub];
            if (!(nelem > 0 && (unsigned long long)elem >= (unsigned long long)nelem)) {
                const double px = pts[i * 3 + 0], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
                load_corners<EXODUS>(conn, nodes, elem, c);
                if (newton_hex8(px, py, pz, c.x, c.y, c.z, xi) && in_hull(xi)) {
                    ok = true;
                    worst = max_abs3(xi);
                }
            }
        }
        // the first candidate, in list order, accepted below 1.025
        const unsigned long long acc = __ballot(ok && worst < (1 + 0.025));
        const unsigned long long mine = G == 64 ? acc : (acc >> (lane & 32)) & 0xffffffffull;
        bool failed = false;
        if (live) {
            double wt[8];
            if (mine) {
                if (sub == __ffsll((long long)mine) - 1) {
                    weights_hex8(xi, wt);
                    emit_row(em, i, c.id, wt);
                }
            } else {
                // the first candidate of smallest max |xi| (the loop: `worst < smallest`, strictly)
                double bw = worst;
                int bj = sub;
#pragma unroll
                for (int off = G / 2; off >= 1; off >>= 1) {
                    const double ow = __shfl_xor(bw, off);
                    const int oj = __shfl_xor(bj, off);
                    if (ow < bw || (ow == bw && oj < bj)) {
                        bw = ow;
                        bj = oj;
                    }
                }
                if (bw < 1.5) {
                    if (sub == bj) {
                        weights_hex8(xi, wt);
                        emit_row(em, i, c.id, wt);
                    }
                } else if (sub == 0) {
                    failed = true;
                    if (zero_failed) {
                        const i64 zid[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        const double zw[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                        emit_row(em, i, zid, zw);
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(failed);
        if (lane == 0 && mask) atomicAdd(nfailed, (unsigned long long)__popcll(mask));
    }
}

#ifndef MM_PASS_ITERS   // tuning builds only
#define MM_PASS_ITERS 6
#endif
#ifndef MM_MID_ITERS
#define MM_MID_ITERS 9
#endif
// Newton caps of the three tiers of solves (loop trips; a solve that converges after n updates needs
// n + 1 trips, the last one only evaluates the residual).  Measured on the metric meshes with the
// reference's arithmetic: 99.99 % of the ACCEPTED solves take 3 or 4 updates, rejected candidates
// mostly 4, 1.6 % five, 0.3 % six to thirteen, and 0.6 % never converge (the reference gives those
// its full 50).  A wave runs in lock step, so every round costs the trips of its slowest lane:
// with one cap of 10 for everybody 42 % of the rounds ran to 10 for the sake of one lane (6.5 trips
// per round on average against 5 for the mean solve).
constexpr int kPassIters = MM_PASS_ITERS;   // ordinary solves: fresh targets and retries
constexpr int kMidIters = MM_MID_ITERS;     // solves that outlast kPassIters, in each other's company
constexpr int kRefIters = 50;               // the reference's own cap (trilinearinterpolator.c:264)


// One pass over all targets (see "Scheduling" in the header comment).
//
// Persistent waves with private queues.  Re-queueing every unresolved target through a global
// queue costs twice: a single device-scope counter serves only ~88 returning atomics per microsecond
// (MI355X_MICROARCH.md, "dequeue"), and -- measured with TCC_MISS -- a later pass over the sparse
// survivors (a third of the targets after the first solve) misses ~7.6 cache lines per solve against
// 4.2 in the first pass, because it touches nearly every line of the candidate rows, points and
// mesh again for a fraction of the work.  So a wave keeps its unresolved targets in LDS and retries
// them itself as soon as it has a full wave of them, and drains the remainder with a few partly
// filled rounds once its input is exhausted.  A solve that outlasts its cap is not abandoned to
// another launch either (three nearly empty persistent launches used to cost 0.3 ms): the target
// waits, same candidate, in the wave's queue of the next tier -- kPassIters -> kMidIters -> the
// reference's 50 -- and is solved again from xi = 0 (the iteration is deterministic: same iterates,
// same verdict) in a round where slow solves only keep each other company.  "Not converged" under
// the reference's cap rejects the candidate, as in the reference.
constexpr int kPassBlock = 256;
constexpr int kGroupListMax = 1 << 16;   // reference-order lists up to this long: one candidate per lane (u^2.2 graded mesh, ~0.5 M on the list: loop 2.9 ms, groups 3.8; u^1.5, 30 k: 1.0 -> 0.1)
#ifndef MM_PASS_WAVES   // tuning builds only: minimum waves per SIMD the register allocator must leave room for
#define MM_PASS_WAVES 2
#endif
#ifndef MM_PASS_PREFETCH   // 1: a fresh batch's targets, first candidates and their connectivity rows are requested a
#define MM_PASS_PREFETCH 0  //    round ahead (measured: no faster -- the pass is bound by its L1 misses in flight, not by the chain)
#endif
#ifndef MM_DRAIN_SLOW_FIRST   // at the end of a wave's input: slow tiers' leftovers before the ordinary drain rounds
#define MM_DRAIN_SLOW_FIRST 0
#endif
#ifndef MM_DRAIN_G8           // drain rounds with at most this many targets try 8 candidates per target at once (0: never)
#define MM_DRAIN_G8 0
#endif
#ifndef MM_FAST_WAVES   // ... for the MM_FP_TOL instances
#define MM_FAST_WAVES 2
#endif
// LDS entries per wave and queue.  A full queue (>= 64 waiting) is served before anything is added to it, the
// slowest tier first: tiers 1 and 2 only grow in rounds of the tier below, which run while they hold fewer
// than 64 (< 128 after the round); tier 0 grows in every round -- 63 + 64 from its own rounds, then one
// round each of tiers 1 and 2 before it is served again: < 256.
// Batches of one plane per panel of the walk order (see the kernel; MM_LOCATE_PANEL overrides, 0 = in order).  -1: as many
// as the XCD has waves in the launch -- wave w then takes column w of a panel in plane after plane: its consecutive
// batches are neighbours ACROSS planes (the same strip of cells, one plane on), which share a layer of elements, and the
// front of all waves is one plane's slice of the panel.  Measured on the metric meshes (locate pass, ms): in order 1.536,
// panels of 64 / 128 / 192 / 384: 1.52 / 1.53 / 1.53 / 1.54, of 256 (= the waves of an XCD) / 512 / 1024 / 2048:
// 1.48 / 1.49 / 1.49 / 1.49.
constexpr int kPassPanel = -1;
constexpr int kWaveQueue0 = 256;
constexpr int kWaveQueue = 128;

// LDS accesses of one wave are served in program order; only the compiler has to be kept from
// moving them across the hand-over points of the wave's queue.
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// Diagnostic builds only (make EXTRA=-DMM_LOCATE_STAMPS): where a wave of locate_pass_kernel spends its cycles
// (s_memtime ticks per phase, summed per workgroup slot; tools/locate_stamps.py prints the shares).
#ifdef MM_LOCATE_STAMPS
constexpr int kLocStampSlots = 4096;
__device__ unsigned long long g_loc_stamps[kLocStampSlots * 8];
#define MM_LSTAMP(n)                                                                   \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        stamp_sum[n] += now_ - stamp_last;                                             \
        stamp_last = now_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#else
#define MM_LSTAMP(n) do { } while (0)
#endif

// Diagnostic builds only (make EXTRA=-DMM_LOCATE_COUNT, printed with MM_LOCATE_DEBUG=1): rounds, lanes and solves of
// the pass kernel, in the first words of the counter block whose word 15 is slow_count.
#ifdef MM_LOCATE_COUNT
#define MM_LCOUNT(slot, pred) do { if (pred) atomicAdd(slow_count - 15 + (slot), 1); } while (0)
#else
#define MM_LCOUNT(slot, pred) do { } while (0)
#endif

// SORTED: the targets come as cell-sorted records {x, y, z, index} (the kNN stage's) and the candidate rows in
// the same order: a wave's 64 targets are 8 neighbouring grid cells -- coordinates and rows stream, and the
// lanes share their candidate elements' connectivity rows and nodes.  i is then the position in that order
// (points, rows, queues), the target's own index (outputs, reference-order list) comes out of its record.
// NID: the type node ids are held in across a solve -- int when the caller knows the mesh has fewer than 2^31 nodes
// (the fused pipeline does), i64 as the connectivity array stores them otherwise.
// FAST (MM_FP_TOL, mm_set_fp_mode): tier 0's solves are newton_hex8_fast under tier 0's cap -- a quarter of the
// instructions, the reference's verdict or "unsure" (mm_newton_hex8.h).  An unsure solve waits, same candidate, in the
// wave's tier-1 queue like a slow one and is solved again there FROM xi = 0 IN THE REFERENCE'S ARITHMETIC (cap kMidIters,
// then tier 2 as ever); what the exact tiers accept leaves with the reference's weights, bit for bit.  (First version:
// a global redo list and a second launch of the exact kernel over it -- 0.35 ms for 1.3 % of the solves: a wave's few
// rounds of 50-trip stragglers and the candidates behind them, one after the other with nothing to overlap them.  Inside
// the one launch they ride along.)  unsure_count (nullable): solves repeated, one atomic per wave at its end.
template <bool EXODUS, typename IDX, bool SORTED, typename NID, bool FAST = false>
__global__ __launch_bounds__(kPassBlock, FAST ? MM_FAST_WAVES : MM_PASS_WAVES) void locate_pass_kernel(i64 k, i64 npoints,
                                                                 const IDX *__restrict__ nn,
                                                                 const i64 *__restrict__ conn, i64 nelem,
                                                                 Emit em,
                                                                 const double *__restrict__ nodes,
                                                                 const double *__restrict__ pts,
                                                                 int *__restrict__ slow_list,
                                                                 int *__restrict__ slow_count,
                                                                 const int *__restrict__ in_list,
                                                                 const int *__restrict__ in_count, int j0,
                                                                 int planes = 0, int panel = 0,
                                                                 int *__restrict__ unsure_count = nullptr,
                                                                 const int *__restrict__ abort6 = nullptr)
{
    if (mm_aborted(abort6)) return;   // (a guessed grid that is not this call's: the host runs the call again)
    // in_list (nullable): only the targets in_list[0 .. *in_count), each from candidate j0 on -- the second
    // pass over the targets that exhausted their lazily evaluated candidates, on their full lists (nn then holds k = the
    // full length per row; candidates before j0 were rejected by the first pass and would be rejected again).
    if (in_list) npoints = (i64)*in_count;
    // [wave][tier][entry]: tier 0 ordinary retries, 1 solves that outlasted kPassIters, 2 ... kMidIters
    // (FAST: tier 1 holds the unsure solves as well, which start again from xi = 0: its iterate queue is not used)
    __shared__ int2 s_queue0[kPassBlock / 64][kWaveQueue0];
    __shared__ int2 s_queue12[kPassBlock / 64][2][kWaveQueue];
    __shared__ double s_qxi[kPassBlock / 64][2][3][kWaveQueue];   // ... and the iterate their solve stopped at
    const int lane = threadIdx.x & 63;
    int2 *const my_q0 = s_queue0[threadIdx.x >> 6];
    int2 *const my_q1 = s_queue12[threadIdx.x >> 6][0];
    int2 *const my_q2 = s_queue12[threadIdx.x >> 6][1];
    double (*const my_xi1)[kWaveQueue] = s_qxi[threadIdx.x >> 6][0];
    double (*const my_xi2)[kWaveQueue] = s_qxi[threadIdx.x >> 6][1];
    int unsure = 0;   // (FAST; wave-uniform) solves handed to the exact tiers
    int held0 = 0, held1 = 0, held2 = 0;   // wave-uniform: entries waiting in each tier's queue

    // XCD-aware deal of the fresh batches.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
    // b + 8 share one) and each XCD has its own 4 MiB L2: with batches dealt round-robin over ALL waves every
    // XCD walks the whole mesh and the connectivity and node arrays cross the fabric eight times (measured:
    // 5-10 GB fetched per pass for 0.9 GB of mesh).  Here XCD x takes the x-th eighth of the targets (mesh
    // nodes come in spatial order: a slab of the domain), its waves interleaved inside that range.
    const int nx = gridDim.x < 8 ? (int)gridDim.x : 8;   // (small launches: one range per workgroup)
    const int xcd = blockIdx.x % nx;
    const i64 nbatches = (npoints + 63) / 64;
    const i64 b_lo = nbatches * xcd / nx, b_hi = nbatches * (xcd + 1) / nx;
    const i64 total = b_hi * 64 < npoints ? b_hi * 64 : npoints;       // end of this XCD's range
    const i64 nwaves = (i64)((gridDim.x - xcd + nx - 1) / nx) * (kPassBlock / 64);   // waves of this XCD
    const i64 wave = (i64)(blockIdx.x / nx) * (kPassBlock / 64) + (threadIdx.x >> 6);
    // Order of the fresh batches inside the XCD's range (SORTED only; planes = x-planes of cells the whole sorted order
    // spans, panel > 0).  In sorted order the front of the XCD's waves sweeps one x-plane of cells after the other, and
    // the connectivity rows and nodes shared with the NEXT plane have left the 4 MiB L2 when that plane comes round
    // (a plane of the metric mesh touches ~9 MB of mesh: every line crossed the fabric 2.2 times).  So the range is
    // read as `rows` planes of `cols` batches each and walked panel by panel: `panel` neighbouring batches of every
    // plane in turn, then the next panel -- the front then spans a few planes of one narrow strip, whose mesh lines
    // stay resident.  Any bijection of the batches is correct; when the planes hold unequal numbers of targets the
    // rows are only roughly the planes and the order is merely less local.  seq -> batch below; sequence numbers
    // that fall outside the range (the last panel, the last row) are skipped.
    // (everything here is wave-uniform and fits 32 bits: npoints < 2^31; kept in scalar registers)
    const int nb_x = (int)(b_hi - b_lo);
    int rows = 1, cols = nb_x, seq_end = nb_x;
    if (panel < 0) panel = (int)nwaves;   // the default: see kPassPanel
    if (SORTED && planes > 0 && panel > 0 && !in_list) {
        rows = (int)(((i64)planes * nb_x + nbatches / 2) / nbatches);
        rows = rows < 1 ? 1 : rows;
        cols = (nb_x + rows - 1) / rows;
        const i64 se = (i64)((cols + panel - 1) / panel) * panel * rows;
        if (se < (i64)0x7fffffff - 4 * (i64)nwaves) seq_end = (int)se;
        else rows = 1;   // (never: < 2^25 batches)
    }
    const int seq_step = (int)nwaves;
    int seq = __builtin_amdgcn_readfirstlane((int)wave);   // this wave's next sequence number
    i64 next = total;  // first target of this wave's next fresh batch (total: none left)
    auto advance = [&]() {
        next = total;
        for (; seq < seq_end; seq += seq_step) {
            int b = seq;
            if (rows > 1) {
                const unsigned per_panel = (unsigned)rows * (unsigned)panel;
                const unsigned pnl = (unsigned)seq / per_panel, rem = (unsigned)seq - pnl * per_panel;
                const unsigned row = rem / (unsigned)panel, col = pnl * (unsigned)panel + (rem - row * (unsigned)panel);
                if (col >= (unsigned)cols) continue;
                b = (int)(row * (unsigned)cols + col);
            }
            if (b < nb_x) {
                next = (b_lo + b) * 64;
                seq += seq_step;
                break;
            }
        }
    };
    advance();
    // A ROUND AHEAD (MM_PASS_PREFETCH): a fresh batch needs its targets' records, their first candidates, those elements'
    // connectivity rows and then the corner coordinates -- four dependent round trips before the first multiplication,
    // and with the cheap MM_FP_TOL solve the pass waits for memory 70 % of its wave-cycles.  The batch a wave will
    // take next is known as soon as it takes one, so its records and first candidates are requested then (stage a) and
    // the connectivity rows at the end of that round (stage b: the candidates have arrived by then); they land while
    // the wave works on other rounds and wait in 17 registers: a fresh round starts at the corner coordinates.
    const bool pf_on = MM_PASS_PREFETCH && !in_list;
    double pf_p[4] = {0., 0., 0., 0.};
    int pf_elem = -1;
    NID pf_id[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool need_b = false;   // (wave-uniform)
    auto prefetch_a = [&]() {
        const i64 q = next + lane;
        if (q < total) {   // (next == total: no batch left, nothing is requested)
            if (SORTED) {
                const double2 *r2 = reinterpret_cast<const double2 *>(pts + q * 4);
                const double2 xy = r2[0], zw = r2[1];
                pf_p[0] = xy.x;
                pf_p[1] = xy.y;
                pf_p[2] = zw.x;
                pf_p[3] = zw.y;
            } else {
                pf_p[0] = pts[q * 3 + 0];
                pf_p[1] = pts[q * 3 + 1];
                pf_p[2] = pts[q * 3 + 2];
            }
            pf_elem = j0 < k ? (int)nn[q * k + j0] : -1;
        }
    };
    auto prefetch_b = [&]() {
        const i64 q = next + lane;
        if (q < total && pf_elem >= 0 && !(nelem > 0 && (i64)pf_elem >= nelem)) load_ids<EXODUS>(conn, (i64)pf_elem, pf_id);
    };
    if (pf_on) {
        prefetch_a();
        prefetch_b();
    }
#ifdef MM_LOCATE_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (;;) {
        // A wave always runs DENSE: as soon as 64 entries are waiting in a queue it solves those (most
        // recent first: their point, candidate row and mesh lines are still in cache), the slower tiers
        // first; otherwise it takes 64 fresh targets.
        bool active;
        i64 i = 0;
        int j = 0;
        double xi_in[3] = {0., 0., 0.};   // tiers 1 and 2: the iterate the solve stopped at under the tier below's cap
        int tier = 0;          // whose cap this round's solves run under
        int lgG = 0;           // log2 of the lanes per target (only the drain rounds work ahead)
        bool fresh = false;    // this round's targets and first candidates wait in the prefetch registers
        double cur_p[4] = {0., 0., 0., 0.};
        int cur_elem = -1;
        NID cur_id[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (held2 >= 64 || held1 >= 64 || held0 >= 64) {
            int from;
            if (held2 >= 64) {
                tier = 2;
                from = held2 -= 64;
            } else if (held1 >= 64) {
                tier = 1;
                from = held1 -= 64;
            } else {
                from = held0 -= 64;
            }
            const int2 e = (tier == 0 ? my_q0 : (tier == 1 ? my_q1 : my_q2))[from + lane];
            i = e.x;
            j = e.y;
            active = true;
            if (tier > 0) {
                double (*const qx)[kWaveQueue] = tier == 1 ? my_xi1 : my_xi2;
                xi_in[0] = qx[0][from + lane];
                xi_in[1] = qx[1][from + lane];
                xi_in[2] = qx[2][from + lane];
            }
        } else if (next < total) {
            const i64 q = next + lane;
            active = q < total;
            j = j0;
            if (active) {
                i = in_list ? (i64)in_list[q] : q;
            }
            if (pf_on) {
                fresh = true;
#pragma unroll
                for (int a = 0; a < 4; ++a) cur_p[a] = pf_p[a];
                cur_elem = pf_elem;
#pragma unroll
                for (int n = 0; n < 8; ++n) cur_id[n] = pf_id[n];
            }
            advance();
            if (pf_on) {
                prefetch_a();
                need_b = true;
            }
#ifdef MM_EXP_NODRAIN   // timing experiment only (results are wrong): how long the drain at the end of the pass takes
        } else if (true) {
            break;
#endif
        } else if ((held1 > 0 || held2 > 0) && (MM_DRAIN_SLOW_FIRST || held0 == 0)) {
            // input exhausted: the slow solves that are left, one lane each -- the slowest tier first, so that what they
            // reject joins the drain rounds below instead of starting another chain of rounds behind them
            tier = MM_DRAIN_SLOW_FIRST ? (held2 > 0 ? 2 : 1) : (held1 > 0 ? 1 : 2);
            active = lane < (tier == 1 ? held1 : held2);
            if (active) {
                const int2 e = (tier == 1 ? my_q1 : my_q2)[lane];
                i = e.x;
                j = e.y;
                double (*const qx)[kWaveQueue] = tier == 1 ? my_xi1 : my_xi2;
                xi_in[0] = qx[0][lane];
                xi_in[1] = qx[1][lane];
                xi_in[2] = qx[2][lane];
            }
            if (tier == 1) held1 = 0;
            else held2 = 0;
        } else if (held0 > 0) {
            // ... then what is left of the ordinary retries, with partly filled waves (63 -> ~22 -> ~8 -> ...: a
            // handful of short rounds at the very end of the pass instead of another pass).  The idle
            // lanes work ahead: with at most 32 (16, 8) targets left, 2 (4, 8) lanes per target try its
            // next 2 (4, 8) candidates at once, which shortens the chain of rounds a hard target needs;
            // the group then acts on the first candidate, in order, that is not a plain rejection.
            lgG = held0 <= MM_DRAIN_G8 ? 3 : (held0 <= 16 ? 2 : (held0 <= 32 ? 1 : 0));
            const int entry = lane >> lgG;
            active = entry < held0;
            if (active) {
                const int2 e = my_q0[entry];
                i = e.x;
                j = e.y + (lane & ((1 << lgG) - 1));
            }
            held0 = 0;
        } else {
            break;
        }
        const int cap = tier == 0 ? kPassIters : (tier == 1 ? kMidIters : kRefIters);
        MM_LCOUNT(0, lane == 0);
        MM_LCOUNT(1, active);
        MM_LCOUNT(6, lane == 0 && tier > 0);
        wave_fence();  // the queue reads above happen before this round's appends
        // outcome of this lane's candidate: 0 rejected, 1 accepted, 2 too slow for this cap (next tier),
        // 3 no candidate left
        int outcome = 0;
        CornersT<NID> c;
        double wt[8];
        MM_LSTAMP(0);   // round selection, queue reads
        i64 tid = i;   // the target's own index
        if (active) {
            double px, py, pz;
            if (fresh) {
                px = cur_p[0];
                py = cur_p[1];
                pz = cur_p[2];
                if (SORTED) tid = (i64)(int)__double_as_longlong(cur_p[3]);
            } else if (SORTED) {
                const double2 *r2 = reinterpret_cast<const double2 *>(pts + i * 4);
                const double2 xy = r2[0], zw = r2[1];
                px = xy.x;
                py = xy.y;
                pz = zw.x;
                tid = (i64)(int)__double_as_longlong(zw.y);
            } else {
                px = pts[i * 3 + 0];
                py = pts[i * 3 + 1];
                pz = pts[i * 3 + 2];
            }
#ifdef MM_LOCATE_STAMPS
            asm volatile("" ::"v"(px), "v"(py), "v"(pz));
            MM_LSTAMP(1);   // the target's coordinates here
#endif
            // skip-scan: next candidate whose x/y corner box (widened by 5 %) contains the point (a lane
            // that works ahead looks at its one candidate only: outside the box = rejected)
            bool have = false;
            for (; j < k; ++j) {
                const bool from_pf = fresh && j == j0;
                const i64 elem = from_pf ? (i64)cur_elem : (i64)nn[i * k + j];
                const bool valid_elem = !(nelem > 0 && (unsigned long long)elem >= (unsigned long long)nelem);
                bool outside = true;
                if (valid_elem) {
                    if (from_pf) {
#pragma unroll
                        for (int n = 0; n < 8; ++n) c.id[n] = cur_id[n];
                    } else {
                        load_ids<EXODUS>(conn, elem, c.id);
                    }
                    load_xyz(nodes, c);
                    double xlo = c.x[0], xhi = c.x[0], ylo = c.y[0], yhi = c.y[0];
#pragma unroll
                    for (int n = 1; n < 8; ++n) {
                        xlo = fmin(xlo, c.x[n]);
                        xhi = fmax(xhi, c.x[n]);
                        ylo = fmin(ylo, c.y[n]);
                        yhi = fmax(yhi, c.y[n]);
                    }
                    const double mx = 0.05 * (xhi - xlo) + 1e-7 * fmax(xhi - xlo, yhi - ylo);
                    const double my = 0.05 * (yhi - ylo) + 1e-7 * fmax(xhi - xlo, yhi - ylo);
                    // NaN corners or point: comparisons are false -> treated as "inside" (never skipped)
                    outside = px < xlo - mx || px > xhi + mx || py < ylo - my || py > yhi + my;
#ifdef MM_EXP_ZBOX   // experiment only (NOT the reference's semantics: it never tests the z residual)
                    {
                        double zlo = c.z[0], zhi = c.z[0];
#pragma unroll
                        for (int n = 1; n < 8; ++n) {
                            zlo = fmin(zlo, c.z[n]);
                            zhi = fmax(zhi, c.z[n]);
                        }
                        const double mz = (0.01 * MM_EXP_ZBOX) * (zhi - zlo);
                        outside = outside || pz < zlo - mz || pz > zhi + mz;
                    }
#endif
                }
                MM_LCOUNT(4, outside && valid_elem);   // candidates dropped by the box test (their rows and corners were loaded)
                if (!outside) {
                    have = true;
                    break;
                }
                if (lgG > 0) break;  // working ahead: this one candidate only
            }
#ifdef MM_LOCATE_STAMPS
            asm volatile("" ::"v"(c.x[0]), "v"(c.z[7]));
            MM_LSTAMP(2);   // candidate row, connectivity row, corner coordinates, box test
#endif
            MM_LCOUNT(2, have && j < k);
            if (j >= k) {
                outcome = 3;  // no candidate left that could be accepted: fallback / failure is the reference's call
            } else if (FAST && tier == 0 && have) {
                double xi[3];
                const int verdict = newton_hex8_fast(px, py, pz, c.x, c.y, c.z, xi, kPassIters);
                if (verdict == MM_FAST_ACCEPT) {
                    weights_hex8_fast(xi, wt);
                    outcome = 1;
                } else if (verdict == MM_FAST_UNSURE) {
                    outcome = 2;
                }
            } else if (have) {
                double xi[3] = {xi_in[0], xi_in[1], xi_in[2]};
                // (a tier's solves start where the tier below's cap stopped them: trips [first, cap); FAST: tier 1 repeats
                // tier 0's fast solve in the reference's arithmetic, from the start)
                const int first = tier == 0 ? 0 : (tier == 1 ? (FAST ? 0 : kPassIters) : kMidIters);
                const bool converged = newton_hex8(px, py, pz, c.x, c.y, c.z, xi, cap, first);
                xi_in[0] = xi[0];
                xi_in[1] = xi[1];
                xi_in[2] = xi[2];
#ifdef MM_LOCATE_STAMPS
                asm volatile("" ::"v"(xi[0]), "v"(xi[2]));
                MM_LSTAMP(3);   // Newton
#endif
                if (converged && in_hull(xi) && max_abs3(xi) < (1 + 0.025)) {
                    weights_hex8(xi, wt);
                    outcome = 1;
                } else if (!converged && cap < kRefIters) {
                    outcome = 2;
                }
            }
        }
        MM_LCOUNT(3, active && outcome == 1);
        MM_LCOUNT(5, active && outcome == 2);
        MM_LSTAMP(4);   // weights
        // the first lane of a target's group (the whole group when nobody works ahead) whose outcome is
        // not a rejection decides; if all rejected, the group's first lane moves on behind the group
        bool requeue = false, slower = false;
        int requeue_j = 0;
        {
            const int G = 1 << lgG;
            const unsigned long long decisive = __ballot(active && outcome != 0);
            const int group_base = lane & ~(G - 1);
            const unsigned long long mine = (decisive >> group_base) & ((1ull << G) - 1ull);
            const int first = mine ? __ffsll((long long)mine) - 1 : G;
            const int g = lane - group_base;
            if (active && g == first) {
                if (outcome == 1) {
                    if (FAST && tier == 0) emit_row<true>(em, tid, c.id, wt);
                    else emit_row<false>(em, tid, c.id, wt);
                }
                else if (outcome == 2) slower = true;
                else slow_list[atomicAdd(slow_count, 1)] = (int)tid;
            } else if (active && first == G && g == 0) {
                // every candidate of the group rejected (j is this lane's, the group's first)
                if (j + G < k) {
                    requeue = true;
                    requeue_j = j + G;
                } else {
                    slow_list[atomicAdd(slow_count, 1)] = (int)tid;
                }
            }
        }
        MM_LSTAMP(5);   // emit: gathers of the field, stores
        // unresolved targets go to the wave's own queues (slot = ballot prefix, no atomics); fewer than
        // 64 were waiting in a queue and at most 64 are added, so kWaveQueue = 128 entries suffice
        const unsigned long long vote = __ballot(requeue);
        if (requeue) my_q0[held0 + __popcll(vote & ((1ull << lane) - 1ull))] = make_int2((int)i, requeue_j);
        held0 += __popcll(vote);
        {
            // same candidate again, under the next tier's cap (a drain round with work-ahead groups runs
            // under tier 0's cap, so its slow solves go to tier 1 like everybody else's)
            const int up = tier == 0 ? 1 : 2;
            const unsigned long long svote = __ballot(slower);
            if (slower) {
                const int at = (up == 1 ? held1 : held2) + __popcll(svote & ((1ull << lane) - 1ull));
                (up == 1 ? my_q1 : my_q2)[at] = make_int2((int)i, j);
                double (*const qx)[kWaveQueue] = up == 1 ? my_xi1 : my_xi2;
                qx[0][at] = xi_in[0];
                qx[1][at] = xi_in[1];
                qx[2][at] = xi_in[2];
            }
            if (FAST && tier == 0) unsure += __popcll(svote);
            if (up == 1) held1 += __popcll(svote);
            else held2 += __popcll(svote);
        }
        wave_fence();
        if (need_b) {
            prefetch_b();
            need_b = false;
        }
        MM_LSTAMP(6);   // queue appends
    }
    if (FAST && unsure_count && lane == 0 && unsure > 0) atomicAdd(unsure_count, unsure);
#ifdef MM_LOCATE_STAMPS
    if (lane == 0) {
        unsigned long long *slot = g_loc_stamps + (size_t)((blockIdx.x * (kPassBlock / 64) + (threadIdx.x >> 6)) & (kLocStampSlots - 1)) * 8;
        for (int q = 0; q < 7; ++q) slot[q] += stamp_sum[q];
        slot[7] += 1ull;
    }
#endif
}

}  // namespace

// The instance of the pass kernel for a launch (all instances share one signature).
template <typename IDX>
struct PassFn {
    typedef void (*type)(i64, i64, const IDX *, const i64 *, i64, Emit, const double *, const double *, int *, int *,
                         const int *, const int *, int, int, int, int *, const int *);
};

template <typename IDX, bool FAST>
static typename PassFn<IDX>::type pass_kernel_for(bool exodus, bool sorted, bool nid32)
{
#define MM_PASS_PICK(EX, SO) (nid32 ? locate_pass_kernel<EX, IDX, SO, int, FAST> : locate_pass_kernel<EX, IDX, SO, i64, FAST>)
    if (sorted) return exodus ? MM_PASS_PICK(true, true) : MM_PASS_PICK(false, true);
    return exodus ? MM_PASS_PICK(true, false) : MM_PASS_PICK(false, false);
#undef MM_PASS_PICK
}

// workgroups of `fn` the device keeps resident (the pass kernel is launched with exactly that many)
template <typename FN>
static i64 resident_workgroups(mm_context *ctx, FN fn)
{
    int per_cu = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kPassBlock, 0);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    if (e != hipSuccess || per_cu < 1 || cus < 1) {
        (void)hipGetLastError();
        per_cu = 2;
        cus = 256;
    }
    return (i64)per_cu * cus;
}

// Launch the whole locate stage on ctx->stream (no synchronisation).  Scratch: the long queue,
// the reference-order list and their counters come from the context's scratch pool, so this must be the only
// scratch user between mm_scratch_begin calls of the caller -- it calls mm_scratch_begin itself.
template <typename IDX>
static int launch_locate_typed(mm_context *ctx, i64 k, i64 npoints, const IDX *nn, const i64 *conn, i64 nelem,
                               int conn_is_exodus, const Emit &em, const double *nodes, const double *pts,
                               i64 *d_nfailed, int zero_failed, const mm_lazy_lists *lazy, const double *tsorted)
{
    // the failed-point counter and the stage's own 16 counters ([15] length of the reference-order list, [14] of the
    // second pass's, [13] the solves MM_FP_TOL repeated exactly) sit in one block of the context's counter array (mm_common.h): ONE
    // fill clears both
    int *counters = reinterpret_cast<int *>(ctx->d_counters + 8);
    if (d_nfailed == ctx->d_counters) {
        if (mm_zero_async(ctx, ctx->d_counters, 16 * sizeof(i64)) != MM_OK) return MM_ERR_HIP;
    } else {
        MM_HIP_CHECK(hipMemsetAsync(d_nfailed, 0, sizeof(i64), ctx->stream));
        MM_HIP_CHECK(hipMemsetAsync(counters, 0, 16 * sizeof(int), ctx->stream));
    }
    if (npoints == 0 || k == 0) return MM_OK;
    MM_REQUIRE(npoints < (i64)0x7fffffff, "too many targets for one launch");
    const int block = 256;
    const i64 full_grid = (npoints + block - 1) / block;
    const bool fast = ctx->fp_mode == MM_FP_TOL;

    // (a buffer of its own, not the scratch pool: a long list's on-demand neighbour query carves the pool anew)
    char *slow_block = nullptr;
    const size_t list_bytes = mm_round256((size_t)npoints * sizeof(int));
    int rc = mm_buffer_get(ctx, MM_BUF_LOC_SLOW, 2 * list_bytes, (void **)&slow_block);
    if (rc != MM_OK) return rc;
    int *slow = (int *)slow_block;
    int *slow2 = (int *)(slow_block + list_bytes);
    int *slow_count = counters + 15;
    int *unsure_count = counters + 13;
    // (node ids in 32 bits when the caller has told us how many nodes there are: fewer registers across the solve)
    const bool nid32 = em.nnodes > 0 && em.nnodes < (i64)0x7fffffff;
    // (tsorted: rows nn[] and the records are in the kNN stage's cell-sorted order; the reference-order kernel
    // below works on the targets' own indices either way)
    const typename PassFn<IDX>::type first_fn = fast ? pass_kernel_for<IDX, true>(conn_is_exodus != 0, tsorted != nullptr, nid32)
                                                     : pass_kernel_for<IDX, false>(conn_is_exodus != 0, tsorted != nullptr, nid32);
    const i64 resident = resident_workgroups(ctx, first_fn);
    i64 grid = resident < full_grid ? resident : full_grid;
    {
        // (tuning: MM_LOCATE_BATCHES_PER_WAVE = fewest fresh batches a wave should get before the grid is cut down)
        static const i64 min_batches = getenv("MM_LOCATE_BATCHES_PER_WAVE") ? atoll(getenv("MM_LOCATE_BATCHES_PER_WAVE")) : 0;
        if (min_batches > 0) {
            const i64 want = ((npoints + 63) / 64 + min_batches * (kPassBlock / 64) - 1) / (min_batches * (kPassBlock / 64));
            if (want < grid) grid = want < 8 ? 8 : want;
        }
    }
    // ONE launch over all targets.  Persistent waves: exactly as many workgroups as the device keeps
    // resident, so that every wave lives for the whole pass and its private queues see a long stream
    // of targets.
    mm_stage_begin(ctx, MM_STAGE_LOCATE_PASS0);
    {
        dim3 g_((unsigned)grid), b_(block);
        // (walk order of the sorted targets: see the kernel; the kNN grid's x dimension comes with the lazy lists)
        static const int panel_env = getenv("MM_LOCATE_PANEL") ? atoi(getenv("MM_LOCATE_PANEL")) : kPassPanel;
        const int planes = (tsorted && lazy && lazy->index && !lazy->index->graded()) ? lazy->index->dims[0] : 0;
        const int panel = panel_env;   // (0: in order, < 0: the kernel's default)
        hipLaunchKernelGGL(first_fn, g_, b_, 0, ctx->stream, k, npoints, nn, conn, nelem, em, nodes, tsorted ? tsorted : pts,
                           slow, slow_count, (const int *)nullptr, (const int *)nullptr, 0, planes, panel, unsure_count,
                           (const int *)ctx->abort_flags);
    }
    mm_stage_end(ctx, MM_STAGE_LOCATE_PASS0);
    // out of candidates without an acceptance: reference-order kernel
    {
        // lazily evaluated lists: these targets have only seen the nearest k of k_full candidates --
        // fetch their full lists now; the reference-order kernel starts again at candidate 0 anyway
        i64 k_slow = k;
        const IDX *nn_slow = nn;
        if (lazy && sizeof(IDX) == sizeof(int)) {
            // A graded cloud (the index has density levels) leaves long lists here -- elongated elements whose centroid
            // is not among a target's eight nearest --: worth one small readback to send them through the tiled kernels
            // instead of the list-mode ones.  A uniform mesh (no levels) leaves none or a handful: no readback.
            i64 list_len = -1;
            if (lazy->index->graded()) {
                MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters + 1, slow_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                list_len = (i64) * reinterpret_cast<const int *>(ctx->h_counters + 1);
            }
            int lrc = mm_knn_query_list_impl(ctx, lazy->index, pts, npoints, lazy->k_full, lazy->nn_full, slow,
                                             slow_count, list_len);
            if (lrc != MM_OK) return lrc;
            k_slow = lazy->k_full;
            nn_slow = reinterpret_cast<const IDX *>(lazy->nn_full);
            if (list_len >= MM_LONG_LIST_MIN) {
                // a long list: its targets walk the REST of their candidates (from the k-th on) in a second launch of
                // the pass kernel; only what finds no acceptance there either is left to the reference-order kernel
                // (which starts again at candidate 0: smallest-error fallback, failures)
                int *slow2_count = counters + 14;
                i64 g2 = (list_len + kPassBlock - 1) / kPassBlock;
                if (g2 > resident) g2 = resident;
                dim3 g_((unsigned)g2), b_(block);
                const bool nid32b = em.nnodes > 0 && em.nnodes < (i64)0x7fffffff;
                hipLaunchKernelGGL((fast ? pass_kernel_for<IDX, true>(conn_is_exodus != 0, false, nid32b)
                                         : pass_kernel_for<IDX, false>(conn_is_exodus != 0, false, nid32b)),
                                   g_, b_, 0, ctx->stream, k_slow, npoints, nn_slow, conn, nelem, em, nodes, pts, slow2, slow2_count,
                                   (const int *)slow, (const int *)slow_count, (int)k, 0, 0, unsure_count,
                                   (const int *)ctx->abort_flags);
                slow = slow2;
                slow_count = slow2_count;
            }
        }
        i64 sgrid = full_grid >> 3;
        if (sgrid < 256) sgrid = full_grid < 256 ? full_grid : 256;
        dim3 g((unsigned)sgrid), b(block);
        // Two launches, one of which returns at once: lists up to kGroupListMax take one candidate per lane (a target's
        // critical path is ONE solve), longer ones the loop (see locate_hex8_group_kernel).  MM_LOCATE_GROUP=0: the loop only.
        static const int group_max = getenv("MM_LOCATE_GROUP") && atoi(getenv("MM_LOCATE_GROUP")) == 0 ? -1 : kGroupListMax;
        if (group_max >= 0 && k_slow <= 64) {
            i64 ggrid = (npoints * (k_slow <= 32 ? 32 : 64) + 255) / 256;
            if (ggrid > 8192) ggrid = 8192;
            if (ggrid < 1) ggrid = 1;
            dim3 gg((unsigned)ggrid);
#define MM_GROUP(EX, GG)                                                                                                   \
    hipLaunchKernelGGL((locate_hex8_group_kernel<EX, IDX, GG>), gg, b, 0, ctx->stream, k_slow, npoints, nn_slow, conn, nelem, em, \
                       nodes, pts, (unsigned long long *)d_nfailed, slow, slow_count, zero_failed, group_max,                \
                       (const int *)ctx->abort_flags)
            if (conn_is_exodus) {
                if (k_slow <= 32) MM_GROUP(true, 32);
                else MM_GROUP(true, 64);
            } else {
                if (k_slow <= 32) MM_GROUP(false, 32);
                else MM_GROUP(false, 64);
            }
#undef MM_GROUP
        }
        const int loop_min = group_max >= 0 && k_slow <= 64 ? group_max : -1;
        if (conn_is_exodus)
            hipLaunchKernelGGL((locate_hex8_kernel<true, IDX>), g, b, 0, ctx->stream, k_slow, npoints, nn_slow, conn,
                               nelem, em, nodes, pts, (unsigned long long *)d_nfailed, slow, slow_count, zero_failed,
                               (const int *)ctx->abort_flags, loop_min);
        else
            hipLaunchKernelGGL((locate_hex8_kernel<false, IDX>), g, b, 0, ctx->stream, k_slow, npoints, nn_slow, conn,
                               nelem, em, nodes, pts, (unsigned long long *)d_nfailed, slow, slow_count, zero_failed,
                               (const int *)ctx->abort_flags, loop_min);
    }
    MM_HIP_CHECK(hipGetLastError());
    static const bool dbg_locate = getenv("MM_LOCATE_DEBUG") != nullptr;
    if (dbg_locate) {
        // diagnostic: sizes of the pass queues and of the reference-order list (synchronises)
        int h[16];
        MM_HIP_CHECK(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[mm_locate] %lld targets; reference-order list %d\n", (long long)npoints, h[15]);
#ifdef MM_LOCATE_COUNT
        fprintf(stderr, "[mm_locate] rounds %d (slow tiers %d), active lanes %d, solves %d, accepted %d, sent to the next tier %d, "
                        "candidates dropped by the box test %d\n", h[0], h[6], h[1], h[2], h[3], h[5], h[4]);
#endif
    }
    return MM_OK;
}

// nn_is_int32 / zero_failed: the fused pipeline's int32 candidate lists and un-zeroed private
// outputs (failed rows are zeroed by the reference-order kernel, the only place a point can fail).
// enc/w may be null when out is given (values only); fields/out null: operator only.
int mm_launch_locate_hex8(mm_context *ctx, i64 k, i64 npoints, const void *nn, bool nn_is_int32, const i64 *conn,
                          i64 nelem, int conn_is_exodus, i64 *enc, const double *nodes, double *w, const double *pts,
                          i64 *d_nfailed, int zero_failed, const double *fields, i64 nnodes, i64 ncomp, double *out,
                          const mm_lazy_lists *lazy, const double *tsorted)
{
    Emit em;
    em.enc = (enc && w) ? enc : nullptr;
    em.w = (enc && w) ? w : nullptr;
    em.fields = fields;
    em.nnodes = nnodes;
    em.ncomp = (int)ncomp;
    em.out = (fields && out && ncomp > 0) ? out : nullptr;
    if (nn_is_int32)
        return launch_locate_typed<int>(ctx, k, npoints, (const int *)nn, conn, nelem, conn_is_exodus, em, nodes, pts,
                                        d_nfailed, zero_failed, lazy, tsorted);
    return launch_locate_typed<i64>(ctx, k, npoints, (const i64 *)nn, conn, nelem, conn_is_exodus, em, nodes, pts,
                                    d_nfailed, zero_failed, nullptr, nullptr);
}

#ifdef MM_LOCATE_STAMPS
extern "C" int mm_debug_locate_stamps(unsigned long long *out16, int reset)
{
    static unsigned long long host[kLocStampSlots * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_loc_stamps), sizeof(host)) != hipSuccess) return -1;
    for (int q = 0; q < 16; ++q) out16[q] = 0;
    for (int b = 0; b < kLocStampSlots; ++b) {
        for (int q = 0; q < 7; ++q) out16[q] += host[b * 8 + q];
        out16[15] += host[b * 8 + 7];
    }
    if (reset) {
        memset(host, 0, sizeof(host));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_loc_stamps), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int64_t mm_locate_hex8(mm_context *ctx, int64_t k, int64_t npoints, const int64_t *nn_d,
                                  const int64_t *conn_d, int64_t nelem, int conn_is_exodus,
                                  int64_t *enc_d, const double *nodes_d, double *w_d,
                                  const double *pts_d)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(k >= 0 && npoints >= 0, "negative size");
    MM_REQUIRE(npoints == 0 || k == 0 || (nn_d && conn_d && enc_d && nodes_d && w_d && pts_d),
               "null array");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    mm_stage_reset(ctx);
    mm_stage_begin(ctx, MM_STAGE_LOCATE);
    int rc = mm_launch_locate_hex8(ctx, k, npoints, nn_d, false, (const i64 *)conn_d, nelem, conn_is_exodus,
                                   (i64 *)enc_d, nodes_d, w_d, pts_d, ctx->d_counters, 0, nullptr, 0, 0, nullptr, nullptr);
    mm_stage_end(ctx, MM_STAGE_LOCATE);
    if (rc != MM_OK) return rc;
    MM_HIP_CHECK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(i64), hipMemcpyDeviceToHost,
                                ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ctx->h_counters[0];
}
