// Internal declarations shared by the HIP translation units of multi_mesh_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "multimesh_hip.h"

typedef long long i64;

void mm_set_error(int code, const char *fmt, ...);

#define MM_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            mm_set_error(MM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                         hipGetErrorString(_e));                                        \
            return MM_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define MM_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            mm_set_error(MM_ERR_ARG, "%s: %s", __func__, msg);                          \
            return MM_ERR_ARG;                                                          \
        }                                                                               \
    } while (0)

// Grow-only device scratch: bump-allocated within one API call, reset at the start of the
// next.  Growing frees the old block after synchronising the stream (only while warming up).
struct mm_scratch {
    char *base = nullptr;
    size_t capacity = 0;
    size_t used = 0;
    void *guard_piece[256] = {};  // MM_GUARD_ALLOC runs: every carve is an allocation of its own (a query over nine density
                                  // levels carves ~8 arrays per level on top of the shared ones)
    int guard_pieces = 0;
};

// Device allocations of the library go through these two.  With MM_GUARD_ALLOC=1 in the environment (test
// runs: tools/guard_run.sh) every allocation -- and every scratch carve -- is mapped so that it ENDS at the
// end of its mapping with unmapped address space behind it: a kernel that reads or writes past an array
// faults at once instead of now and then (there is no GPU AddressSanitizer on this pool).
hipError_t mm_raw_alloc(int device, void **out, size_t bytes);
hipError_t mm_raw_free(void *ptr);
bool mm_guard_alloc(void);

// Grow-only device buffers the fused pipeline reuses from call to call (hipMalloc/hipFree of
// gigabyte-sized intermediates would otherwise cost milliseconds and a device sync per call).
enum mm_buffer_slot {
    MM_BUF_CENTROID = 0,
    MM_BUF_NN,
    MM_BUF_ENC,
    MM_BUF_W,
    MM_BUF_CELL_START,
    MM_BUF_SORTED_XYZ,
    MM_BUF_NN_FULL,
    MM_BUF_BOX_PARTIAL,
    MM_BUF_TSORTED,                       // targets in cell order {x, y, z, index} (kept for the locate stage)
    MM_BUF_H_NODES,                       // device copies of host arrays (mm_interpolate_hex8_host)
    MM_BUF_H_CONN,
    MM_BUF_H_POINTS,
    MM_BUF_H_FIELDS,
    MM_BUF_H_OUT,
    MM_BUF_L_NN,                          // device copies of the LEGACY symbols' host arrays (centroid, triLinearInterpolator):
    MM_BUF_L_CONN,                        //   grow-only like the rest, so a loop of calls (reference scripts/cli.py:183-195
    MM_BUF_L_ENC,                         //   calls triLinearInterpolator once per GLL point of the element: 125 times)
    MM_BUF_L_NODES,                       //   allocates once
    MM_BUF_L_W,
    MM_BUF_L_PTS,
    MM_BUF_LOC_SLOW,                      // locate stage: list of targets for the reference-order kernel + its counters
    MM_BUF_TREE_KEYS,                     // the kNN tree (graded clouds): Morton keys, records, leaf levels, search table
    MM_BUF_TREE_XYZ,
    MM_BUF_TREE_LEVEL,
    MM_BUF_TREE_COARSE,
    MM_BUF_TREE_DOWN,                     //   ... targets of a query whose first window overflowed the tile (second pass)
    MM_BUF_LEVELS,                        // density levels of the kNN grid: {cell_start, sorted_xyz} per level
    MM_BUF_COUNT = MM_BUF_LEVELS + 2 * 8
};

struct mm_context {
    int device = 0;
    hipStream_t stream = nullptr;
    mm_scratch scratch;
    void *buf_ptr[MM_BUF_COUNT] = {};
    size_t buf_cap[MM_BUF_COUNT] = {};
    // failed-point counter + small readback area (device) and its pinned host mirror
    i64 *d_counters = nullptr;
    i64 *h_counters = nullptr;
    // stage timers
    int profiling = 0;
    int lazy_lists = 1;   // mm_set_lazy_lists
    int fp_mode = 0;      // mm_set_fp_mode (MM_FP_EXACT; MM_FP_MODE=tol in the environment starts contexts in MM_FP_TOL)
    hipEvent_t ev_begin[MM_STAGE_COUNT];
    hipEvent_t ev_end[MM_STAGE_COUNT];
    bool ev_used[MM_STAGE_COUNT];
    bool ev_created = false;
    hipEvent_t ev_misc = nullptr;   // host waits on small readbacks while later work stays queued
    // mm_interpolate_hex8: the bounding box of the previous call's centroids (one level, default density), from which
    // the next call over a source mesh of the same size lays its search grid out without waiting for its own box
    // (mm_knn_build_guessed / mm_knn_guess_confirmed in mm_knn.hip); misses: calls that had to be run again
    struct {
        bool valid = false;
        i64 nsrc = 0;
        double box[6] = {0, 0, 0, 0, 0, 0};
        int stat_shift = -1;
        int misses = 0;
        long long calls_guessed = 0;   // (mm_debug_grid_guess: what the tests look at)
    } grid_guess;
    // kNN queries whose targets fill only part of the grid (knn_query_typed): the verdict of the occupied-strip statistic
    struct {
        bool valid = false;
        i64 npts = 0, ncells = 0;
        bool dense = false;
    } lane_hint;
    int *abort_flags = nullptr;   // non-null during a call over a guessed grid: d_counters + kMmAbortSlot (mm_aborted)
    hipStream_t copy_stream = nullptr;   // host-array entry points: uploads run beside the kernels (created on first use)
    hipEvent_t ev_copy[3] = {nullptr, nullptr, nullptr};
};

// Reserve `total` bytes of scratch for the current call (may reallocate), then carve with
// mm_scratch_take.  All carve sizes are rounded up to 256 B.
int mm_scratch_begin(mm_context *ctx, size_t total);
void *mm_scratch_take(mm_context *ctx, size_t bytes);
static inline size_t mm_round256(size_t b) { return (b + 255) & ~(size_t)255; }
// bytes to clear for a carve of `b` bytes: whole 256-byte units (the carve is rounded up to them, and an odd
// tail costs a second fill dispatch) -- except under MM_GUARD_ALLOC, where the carve ends with its array
static inline size_t mm_fill_span(size_t b) { return mm_guard_alloc() ? b : mm_round256(b); }

// Zero `bytes` bytes of device memory on ctx->stream with a kernel of the library's own (see mm_context.hip: the runtime's
// fill is queued behind a barrier).
int mm_zero_async(mm_context *ctx, void *dst_d, size_t bytes);

// n (<= 64) 64-bit words of device memory into the context's pinned mirror, by a kernel on ctx->stream.
int mm_mirror_async(mm_context *ctx, long long *dst_pinned, const long long *src_d, int n);

// Cached buffer of at least `bytes` for `slot` (grows by reallocation, after a stream sync).
int mm_buffer_get(mm_context *ctx, int slot, size_t bytes, void **out);

// Stage timing helpers: no-ops unless profiling is on.
void mm_stage_reset(mm_context *ctx);
void mm_stage_begin(mm_context *ctx, int stage);
void mm_stage_end(mm_context *ctx, int stage);

// Slots of mm_context::d_counters (64 x i64, zeroed when the context is created) / its pinned mirror h_counters:
//   0 failed / missing points of the call    1 mm_layers' counter    8..15 the hex8 locate stage's 16 counters
//   32..47 the grid statistic                48..53 (h_counters) the sources' bounding box
// ("The last workgroup of a launch finishes the reduction" was tried for the bounding box and the scans and
// measured: on this multi-XCD part the device-scope fence every workgroup needs before it takes its ticket writes
// its XCD's L2 back -- the centroid kernel went from 0.26 to 0.72 ms, a scan's first kernel from 5 to 80 us.)
//   24..26 (six ints) mismatch flags of a guessed grid (mm_aborted)   2..3 scratch of small readbacks
constexpr int kMmStatSlot = 32, kMmBoxSlot = 48, kMmAbortSlot = 24;   // (slots 0..15 are cleared by every locate stage)

// A call over a GUESSED search grid (mm_interpolate_hex8): abort6 = six ints, non-zero when this call's bounding box is
// not the one the grid was laid out from.  The kernels that would be ruinously slow on a foreign grid ask first.
__device__ __forceinline__ bool mm_aborted(const int *__restrict__ abort6)
{
    if (!abort6) return false;
    const int4 a = *reinterpret_cast<const int4 *>(abort6);
    const int2 b = *reinterpret_cast<const int2 *>(abort6 + 4);
    return (a.x | a.y | a.z | a.w | b.x | b.y) != 0;
}

// ---- internal launchers (device pointers, no synchronisation) -------------------------
int mm_launch_centroid(mm_context *ctx, i64 ndim, i64 nelem, i64 nper, const i64 *conn,
                       const double *points, double *out);
// hex8 centroids + per-workgroup bounding boxes partial[nblocks][6] of them (fused pipeline)
int mm_launch_centroid_bbox(mm_context *ctx, i64 nelem, const i64 *conn, const double *points, double *out,
                            double *partial, int nblocks);
// enc/w: operator rows (may be null when out is given); fields [ncomp][nnodes] + out [npoints][ncomp]:
// interpolated values formed at the acceptance point (the gather fused into the locate), or null
// lazy (nullable): nn holds only the k nearest of k_full; targets that exhaust them get their full
// list from the index (written to nn_full int32[npoints][k_full], rows of those targets only) and
// go through the reference-order kernel
struct mm_knn_index;
struct mm_lazy_lists {
    const mm_knn_index *index;
    i64 k_full;
    int *nn_full;
};
int mm_launch_locate_hex8(mm_context *ctx, i64 k, i64 npoints, const void *nn, bool nn_is_int32,
                          const i64 *conn, i64 nelem, int conn_is_exodus, i64 *enc, const double *nodes,
                          double *w, const double *pts, i64 *d_nfailed, int zero_failed,
                          const double *fields, i64 nnodes, i64 ncomp, double *out,
                          const mm_lazy_lists *lazy, const double *tsorted = nullptr);
// full-length int32 lists for a device-side list of targets (generic kernel, rows idx[i*k ...])
// on-demand list queries at least this long (graded meshes) take the tiled kernels; the locate stage then also walks
// the rest of those targets' candidates with its pass kernel before the reference-order kernel sees what is left
#define MM_LONG_LIST_MIN 32768
// list_len_hint: the list's length when the caller has read it back (long lists then take the tiled kernels, which carve
// the context's scratch pool anew), -1 when it is only known on the device
int mm_knn_query_list_impl(mm_context *ctx, const mm_knn_index *ix, const double *pts_d, i64 npts, i64 k,
                           int *idx_d, const int *list, const int *list_count, i64 list_len_hint);
int mm_launch_gather(mm_context *ctx, const double *fields, i64 nsrc, i64 ncomp, const i64 *ids,
                     const double *w, i64 npoints, i64 P, double *out, int out_point_major);

// The density-adaptive part of a kNN index (mm_knn_tree.inc.h): the sources in Morton order.
struct mm_knn_tree {
    double lo[3] = {0, 0, 0};      // corner of the bounding cube
    double size = 1.0;             // its edge
    double scale = 1.0;            // finest cells per unit length
    unsigned long long *keys = nullptr;   // [nsrc] sorted Morton keys
    double *xyz = nullptr;                // [nsrc + 1][4] records in that order
    unsigned char *level = nullptr;       // [nsrc] level of the leaf around each source
    int *coarse = nullptr;                // first source of every level-7 cell
    bool borrowed = false;
};

// kNN search grid over source points (device resident).
struct mm_knn_index {
    i64 nsrc = 0;
    int ndim = 3;
    int dims[3] = {1, 1, 1};       // cells per axis
    double lo[3] = {0, 0, 0};      // bounding-box minimum
    double h[3] = {1, 1, 1};       // cell edge per axis
    double inv_h[3] = {1, 1, 1};
    i64 ncells = 1;
    int *cell_start = nullptr;     // [ncells + 1] exclusive prefix of per-cell counts
    double *sorted_xyz = nullptr;  // [nsrc][4] records {x, y, z, original index bits} in cell order
    bool borrowed = false;         // arrays belong to the context's buffer cache (fused pipeline)
    mm_knn_index *fine = nullptr;  // next density level: a grid over the same sources with smaller cells (owned)
    mm_knn_tree *tree = nullptr;   // graded clouds: the adaptive index that serves them instead of density levels (owned)
    bool graded() const { return fine != nullptr || tree != nullptr; }
};
