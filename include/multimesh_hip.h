/*
 * multimesh_hip.h -- C ABI of multi_mesh_hip.so, the MI355X (gfx950) drop-in for the
 * MultiMesh interpolation hot path: element centroids -> k nearest centroids ->
 * point-in-hex8 location by Newton inversion -> shape-function weighted gather.
 *
 * Plain pointers and sizes only.  Two groups of entry points:
 *
 *  (1) LEGACY symbols with the exact signature and semantics of the reference's own C
 *      library (what reference multi_mesh/helpers.py:43-81 binds).  Host pointers in,
 *      host pointers out; the work runs on the GPU.
 *  (2) mm_* symbols taking DEVICE pointers, for callers that keep meshes resident in HBM
 *      (the benchmark, the multi-GPU driver, repeated queries against one source mesh).
 *
 * Error convention: the reference has no error channel (per-point failures are a count,
 * reference src/trilinearinterpolator.c:133-147).  Here: int / int64 returns are >= 0 on
 * success (a failed-point count where the reference returns one) and NEGATIVE MM_ERR_* on a
 * runtime error; mm_last_error() returns a message.  There is no CPU fallback: without a
 * usable GPU every compute entry point fails with MM_ERR_NODEVICE / MM_ERR_HIP.
 * Nothing is ever printed from device code (the reference's "not any" printf is dropped).
 */
#ifndef MULTIMESH_HIP_H
#define MULTIMESH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_OK 0
#define MM_ERR_ARG (-1)         /* bad argument (null pointer, negative size, k too large ...) */
#define MM_ERR_HIP (-2)         /* a HIP runtime call failed; see mm_last_error() */
#define MM_ERR_NODEVICE (-3)    /* no usable GPU */
#define MM_ERR_ALLOC (-4)       /* device or host allocation failed */
#define MM_ERR_UNSUPPORTED (-5) /* valid request this build does not implement */

#define MM_KNN_MAX_K 64 /* largest nelem_to_search served (reference uses 20, 25, 30) */

/* ------------------------------------------------------------------------------------
 * (1) Legacy symbols -- replace the reference library one for one.
 * ---------------------------------------------------------------------------------- */

/* Replaces reference src/centroid.c:3-9 (bound at helpers.py:43-57).
 * centroid[e][a] = (sum over the element's nodes, in connectivity order) / npointsperelem.
 * connectivity int64[nelem][npointsperelem], points f64[npoints][ndim], centroid
 * f64[nelem][ndim], all C-contiguous host arrays.  ndim in {1,2,3}.  The signature has no
 * error channel: on failure the output is left untouched, a message goes to stderr and
 * mm_last_error()/mm_last_status() report it. */
void centroid(long long ndim, long long nelem, long long npointsperelem,
              long long *connectivity, double *points, double *centroid);

/* Replaces reference src/trilinearinterpolator.c:40-48 (bound at helpers.py:59-81).
 * For each of npoints targets walk its nelem_to_search candidate elements in the given
 * order, Newton-invert the trilinear map, accept the first with max|xi| < 1.025, else after
 * the last candidate fall back to the least-outside one if < 1.5; write the element's 8
 * node ids and 8 weights in place; rows of failed points are left untouched.  Returns the
 * number of failed points (>= 0) or a negative MM_ERR_*.  Host arrays:
 * nearest_element_indices int64[npoints][k], connectivity int64[nelem][8] (locator corner
 * order), enclosing_elem_indices int64[npoints][8] out, nodes f64[nnodes][3], weights
 * f64[npoints][8] out, points f64[npoints][3].  nelem / nnodes are not in the reference
 * signature; they are recovered as max index + 1. */
long long triLinearInterpolator(long long nelem_to_search, long long npoints,
                                long long *nearest_element_indices, long long *connectivity,
                                long long *enclosing_elem_indices, double *nodes,
                                double *weights, double *points);

/* ------------------------------------------------------------------------------------
 * (2) Device-pointer API.
 * ---------------------------------------------------------------------------------- */

typedef struct mm_context mm_context;     /* device, stream, scratch pool, stage timers */
typedef struct mm_knn_index mm_knn_index; /* device-resident search grid over source points */

int mm_device_count(void);
const char *mm_last_error(void); /* thread-local message of the last failure */
int mm_last_status(void);        /* thread-local MM_* code of the last legacy call */

/* hip_stream: a hipStream_t (NULL = the device's default stream).  All work of a context is
 * issued on that stream; entry points that return a count synchronise it. */
int mm_context_create(int device, void *hip_stream, mm_context **out);
void mm_context_destroy(mm_context *ctx);
int mm_synchronize(mm_context *ctx);

/* Resident-array helpers for hosts that have no other device allocator. */
int mm_device_alloc(mm_context *ctx, size_t bytes, void **dptr);
int mm_device_free(mm_context *ctx, void *dptr);
int mm_copy_h2d(mm_context *ctx, void *dst_d, const void *src_h, size_t bytes);
int mm_copy_d2h(mm_context *ctx, void *dst_h, const void *src_d, size_t bytes);
int mm_memset(mm_context *ctx, void *dst_d, int value, size_t bytes);

/* A1 -- element centroids (reference src/centroid.c:3-25), device arrays. */
int mm_centroid(mm_context *ctx, int64_t ndim, int64_t nelem, int64_t npointsperelem,
                const int64_t *connectivity_d, const double *points_d, double *centroid_d);

/* A2 -- exact k nearest neighbours; replaces scipy.spatial.cKDTree(src, balanced_tree=False)
 * + .query(pts, k) at reference scripts/cli.py:66-73.  build = the tree construction,
 * query = .query: idx_d int64[npts][k] ascending by Euclidean distance (squared distance
 * summed axis by axis in fp64, no fused multiply-add; equal distances ordered by index);
 * rows with fewer than k sources are padded with index nsrc (distance inf) like cKDTree.
 * dist_d (nullable) f64[npts][k] receives the distances.  ndim in {1,2,3}; k <= MM_KNN_MAX_K. */
int mm_knn_build(mm_context *ctx, const double *src_d, int64_t nsrc, int64_t ndim,
                 mm_knn_index **out);
int mm_knn_query(mm_context *ctx, const mm_knn_index *index, const double *pts_d, int64_t npts,
                 int64_t k, int64_t *idx_d, double *dist_d);
void mm_knn_destroy(mm_context *ctx, mm_knn_index *index);

/* A4..A8 -- hex8 point location + weights (reference src/trilinearinterpolator.c:40-148),
 * device arrays, same in-place contract as the legacy symbol.  nelem > 0 enables a bounds
 * guard: candidates outside [0, nelem) (cKDTree's padding) count as "not in hull".
 * conn_is_exodus != 0 applies the reference's host-side column reorder
 * (scripts/cli.py:79-81) on the fly, so the caller can pass the mesh's own connectivity. */
int64_t mm_locate_hex8(mm_context *ctx, int64_t nelem_to_search, int64_t npoints,
                       const int64_t *nearest_element_indices_d, const int64_t *connectivity_d,
                       int64_t nelem, int conn_is_exodus, int64_t *enclosing_elem_indices_d,
                       const double *nodes_d, double *weights_d, const double *points_d);

/* A9 -- weighted gather; replaces np.sum(field[enc] * w, axis=1) at reference
 * scripts/cli.py:98-100 (P = 8) and interpolator.py:976 (P = 27, 125), NumPy's summation
 * order.  fields_d f64[ncomp][nsrc] (one contiguous array per parameter, as the reference
 * keeps them); ids_d int64[npoints][P]; w_d f64[npoints][P]; out_d f64[npoints][ncomp] when
 * out_point_major (the layout reference interpolator.py:973-977 returns) else
 * f64[ncomp][npoints].  P <= 128. */
int mm_gather(mm_context *ctx, const double *fields_d, int64_t nsrc, int64_t ncomp,
              const int64_t *ids_d, const double *w_d, int64_t npoints, int64_t P, double *out_d,
              int out_point_major);

/* A10 -- GLL elements (order 1, 2, 4; dim 2, 3): element search + Lagrange coefficients; replaces
 * the per-point loop get_element_weights.check_inside at reference
 * components/interpolator.py:1181-1233 (tolerance and snap_to_nearest as there; 1.03 / 0 gives the
 * layered variant :1271-1297).  The inverse transform and the coefficients come from salvus.fem in
 * the reference (absent): PARITY UNPINNED, numerics defined in mm_locate_gll.hip / the oracle.
 * gll_points_d f64[nelem][P][dim] with P = (order+1)^dim, control node p = i + (order+1) j + ...;
 * nn_d int64[npoints][k]; elem_d int64[npoints] (-1 = not found); coeffs_d f64[npoints][P].
 * Returns the number of points without an element, or a negative MM_ERR_*. */
int64_t mm_locate_gll(mm_context *ctx, int order, int dim, int64_t nelem_to_search, int64_t npoints,
                      const int64_t *nearest_element_indices_d, const double *gll_points_d, int64_t nelem,
                      const double *points_d, double tolerance, int snap_to_nearest, int64_t *elem_d,
                      double *coeffs_d);

/* The other GLL acceptance loop of the reference: _check_if_inside_element + boundary_box_check
 * (components/interpolator.py:1350-1367, :1409-1473; used by gll_2_exodus :274, the layered drivers
 * :543 and the helpers :1523, :1572).  Candidates in order: bounding box of the control nodes first,
 * inside -> inverse transform, accept when every |xi| <= 1.04.  Otherwise the first candidate whose
 * box holds the point, else the one with the nearest control-node mean, is transformed again; NaN
 * or any |xi| >= 1.04 gives the reference's constant xi = (0.645, -0.5, 0.22) (:1468-1471).
 * Arrays as mm_locate_gll; PARITY UNPINNED like it.  Returns the number of points whose final
 * transform was NaN (where the reference raises unless ignore_hard_elements), or a negative MM_ERR_*. */
int64_t mm_locate_gll_bbox(mm_context *ctx, int order, int dim, int64_t nelem_to_search, int64_t npoints,
                           const int64_t *nearest_element_indices_d, const double *gll_points_d,
                           int64_t nelem, const double *points_d, int64_t *elem_d, double *coeffs_d);

/* Element-nodal gather np.sum(coeffs * field[elem_indices], axis=1) (reference interpolator.py:976):
 * fields_d f64[ncomp][nelem][P]; points with elem -1 give 0.  NumPy's summation order. */
int mm_gather_elem(mm_context *ctx, const double *fields_d, int64_t nelem, int64_t ncomp,
                   const int64_t *elem_d, const double *coeffs_d, int64_t npoints, int64_t P, double *out_d,
                   int out_point_major);

/* The GLL form of the whole path on resident arrays: what reference components/interpolator.py:931-977
 * (interpolate_to_points on a GLL mesh) and the core of gll_2_gll (:700-830) compute for an array
 * of points -- centroid of every element's control nodes (NumPy mean(axis=1) order,
 * salvus_mesh_reader.py:99-100), the nelem_to_search nearest centroids, the acceptance loop of
 * :1181-1233 (mm_locate_gll), then np.sum(coeffs * field[elem], axis=1) per component (:976).
 *   gll_points_d f64[nelem][P][dim], points_d f64[npoints][dim], fields_d f64[ncomp][nelem][P],
 *   out_d f64[npoints][ncomp]; points that are not found give +-0.0 like NumPy's field[-1] * 0.
 *   elem_out_d int64[npoints] and coeffs_out_d f64[npoints][P]: both or neither (the operator).
 * Without the operator outputs the sum is formed where a target is accepted (no coefficient array
 * in memory), and the candidate lists are evaluated lazily (mm_set_lazy_lists); results are
 * identical to the staged calls.  Returns the number of points not found, or a negative MM_ERR_*. */
int64_t mm_interpolate_gll(mm_context *ctx, int order, int dim, const double *gll_points_d, int64_t nelem,
                           const double *points_d, int64_t npoints, const double *fields_d, int64_t ncomp,
                           int64_t nelem_to_search, double tolerance, int snap_to_nearest, double *out_d,
                           int64_t *elem_out_d, double *coeffs_out_d);

/* Unique points and the index array that rebuilds the input: np.unique(points, axis=0,
 * return_inverse=True) of reference utils.py:484-488 (get_unique_points, the pre-step of the GLL
 * target flows; scatter-back at components/interpolator.py:823).  points_d f64[npoints][dim];
 * unique_d f64[npoints][dim] (room for the worst case), rows in lexicographic (x, y, z) order;
 * inverse_d int64[npoints] with unique[inverse[i]] == points[i].  -0.0 equals +0.0 as in NumPy (the
 * row kept is the one with the smallest index); NaN coordinates are not supported.
 * Returns the number of unique rows, or a negative MM_ERR_*. */
int64_t mm_unique_points(mm_context *ctx, const double *points_d, int64_t npoints, int64_t dim,
                         double *unique_d, int64_t *inverse_d);

/* The same collapse WITHOUT NumPy's order, for callers that only scatter values back through the inverse -- which is all
 * the reference ever does with get_unique_points (components/interpolator.py:823, :1079-1081): the unique rows come in the
 * order of their first occurrence (unique[inverse[i]] == points[i] as above, -0.0 stored as +0.0).  A hash table instead of
 * a sort: 2-3x faster.  Returns the number of unique rows, or a negative MM_ERR_*. */
int64_t mm_unique_points_any_order(mm_context *ctx, const double *points_d, int64_t npoints, int64_t dim,
                                   double *unique_d, int64_t *inverse_d);

/* Layer-aware GLL drivers (reference components/interpolator.py:1047-1082): the scatter-back
 *     new_field[mask[layer]] = values[inverse].reshape(...)            (:1079-1081)
 * on the device.  values_d f64[nunique][ncomp] (what mm_interpolate_gll returns for the layer's unique target
 * points), inverse_d int64[nmasked * P] (mm_unique_points of the layer's element-nodal target points),
 * elem_ids_d int64[nmasked] (the layer's target elements); out_d f64[ncomp][nelem_out][P] receives
 * out[c][elem_ids[m]][p] = values[inverse[m * P + p]][c]; other rows are left alone. */
int mm_scatter_elements(mm_context *ctx, const double *values_d, int64_t nunique, int64_t ncomp,
                        const int64_t *inverse_d, const int64_t *elem_ids_d, int64_t nmasked, int64_t P,
                        int64_t nelem_out, double *out_d);

/* find_gll_coeffs as query_model / gll_2_gll drive it (reference components/interpolator.py:113, :777): the tree is built
 * over ALL GLL points and the neighbour list of point indices becomes a list of element indices by
 * np.floor(index / P) -- in place on the device (idx_d int64[n]). */
int mm_points_to_elements(mm_context *ctx, int64_t *idx_d, int64_t n, int64_t P);

/* The fluid/solid fix-up of gll_2_gll (reference components/interpolator.py:829-841) on element data
 * values_d / previous_d f64[nelem][ncomp][P]: elements with solid_d[e] == 0 get their previous values back
 * ("values[~solid_elements] = new_values[~solid_elements]"), and so does a solid element whose parameter
 * vs_index is exactly 0.0 at any point.  Returns the number of SOLID elements restored, or a negative MM_ERR_*. */
int64_t mm_fluid_solid_fix(mm_context *ctx, double *values_d, const double *previous_d,
                           const unsigned char *solid_d, int64_t nelem, int64_t ncomp, int64_t P,
                           int64_t vs_index);

/* The whole hot path of reference scripts/cli.py:62-100 on resident arrays:
 * centroid -> search grid -> kNN -> locate -> gather.  connectivity_d is the mesh's own
 * (exodus-order) hex8 connectivity.  enc_d / w_d (nullable) receive the interpolation
 * operator (int64[N][8], f64[N][8]); out_d f64[N][ncomp] (nullable when only the operator is
 * wanted).  Returns the number of failed points or a negative MM_ERR_*. */
int64_t mm_interpolate_hex8(mm_context *ctx, const double *nodes_d, int64_t nnodes,
                            const int64_t *connectivity_d, int64_t nelem, const double *points_d,
                            int64_t npoints, const double *fields_d, int64_t ncomp,
                            int64_t nelem_to_search, double *out_d, int64_t *enc_d, double *w_d);

/* A source mesh kept resident for repeated calls -- what the reference does when it builds its cKDTree once and queries it
 * for each of the 125 GLL points of the target elements or for every time step (scripts/cli.py:141-195):
 * mm_source_create computes the element centroids and the search grid ONCE; mm_interpolate_hex8_on is mm_interpolate_hex8
 * without those two stages (results identical, bit for bit).  nodes_d / connectivity_d are BORROWED: they must stay alive
 * and unchanged until mm_source_destroy. */
typedef struct mm_source mm_source;
int mm_source_create(mm_context *ctx, const double *nodes_d, int64_t nnodes, const int64_t *connectivity_d, int64_t nelem,
                     mm_source **out);
void mm_source_destroy(mm_context *ctx, mm_source *source);
int64_t mm_interpolate_hex8_on(mm_context *ctx, const mm_source *source, const double *points_d, int64_t npoints,
                               const double *fields_d, int64_t ncomp, int64_t nelem_to_search, double *out_d,
                               int64_t *enc_d, double *w_d);

/* The same path fed from HOST arrays -- what the reference's callers hold (NumPy arrays handed to
 * centroid / cKDTree / triLinearInterpolator / np.sum at scripts/cli.py:62-100): nodes f64[nnodes][3],
 * connectivity int64[nelem][8] (exodus order), points f64[npoints][3], fields f64[ncomp][nnodes], in place
 * results out_h f64[npoints][ncomp] and (both or neither) enc_h int64[npoints][8], w_h f64[npoints][8].
 * Uploads run on a second stream beside the kernels of the stage before (mesh -> centroids + grid |
 * targets -> kNN | fields -> locate); the device copies live in the context and are reused by later calls. */
int64_t mm_interpolate_hex8_host(mm_context *ctx, const double *nodes_h, int64_t nnodes,
                                 const int64_t *connectivity_h, int64_t nelem, const double *points_h,
                                 int64_t npoints, const double *fields_h, int64_t ncomp,
                                 int64_t nelem_to_search, double *out_h, int64_t *enc_h, double *w_h);

/* mm_interpolate_hex8 evaluates its candidate lists lazily (default on): the locate stage walks a
 * target's candidates in kNN order and stops at the first acceptance (1.6 candidates per target on
 * mesh-like inputs), so the pipeline first asks the kNN stage for the 8 nearest only and computes
 * the full nelem_to_search list just for the targets that exhaust those 8 (they then go through
 * the reference-order locate from candidate 0).  The k' nearest are the first k' of the k nearest,
 * so every output is bit-identical to the eager evaluation; on = 0 forces the eager one. */
int mm_set_lazy_lists(mm_context *ctx, int on);

/* Floating-point mode of the hex8 locate stage (mm_locate_hex8, mm_interpolate_hex8*, triLinearInterpolator).
 *   MM_FP_EXACT (default): the reference's arithmetic operation for operation (src/trilinearinterpolator.c:150-375)
 *     -- node ids, weights and interpolated values bit-identical to the reference.
 *   MM_FP_TOL: the Newton inversion runs in a cheaper arithmetic (polynomial form of the map, fused multiply-adds,
 *     Cramer's rule; csrc/mm_newton_hex8.h) that must CERTIFY every decision the reference makes -- each residual test
 *     against 1e-8 * scale, the final max|xi| against 1.025 -- with a margin two orders above the rounding differences
 *     between the two arithmetics; a solve it cannot certify is repeated in the reference's arithmetic.  Element / node
 *     ids and the failed count stay bit-identical; weights and values agree with the reference to
 *     max(1e-12, 64 eps |x| / h) (|x| / h: coordinate magnitude over element size; 1e-12 on the BASELINE meshes),
 *     relative to max|weight| = 1 resp. max|field|.
 * The environment variable MM_FP_MODE=tol makes MM_FP_TOL the default of new contexts (how a user of the legacy
 * symbols opts in). */
#define MM_FP_EXACT 0
#define MM_FP_TOL 1
int mm_set_fp_mode(mm_context *ctx, int mode);
int mm_get_fp_mode(mm_context *ctx);
/* out4 = {solves of the last hex8 locate stage that MM_FP_TOL repeated in the reference's arithmetic, targets that went
 * through the reference-order kernel, targets of a long on-demand list's second pass, 0}.  Synchronises. */
int mm_last_locate_stats(mm_context *ctx, long long *out4);

/* Stage timers (hipEvents on the context's stream).  With profiling on, every kernel
 * launched by the calls above is bracketed by events; mm_last_timings fills ms[stage] for
 * the stages of the LAST call (0 for stages that did not run) and returns the stage count. */
enum mm_stage {
    MM_STAGE_CENTROID = 0,
    MM_STAGE_KNN_BUILD = 1,
    MM_STAGE_KNN_QUERY = 2,    /* whole query: target sort + cell kernel + straggler kernel */
    MM_STAGE_LOCATE = 3,       /* whole locate: all launches + reference-order kernel (+ on-demand full lists) */
    MM_STAGE_GATHER = 4,
    MM_STAGE_KNN_CELL = 5,     /* the kNN cell kernel alone (inside MM_STAGE_KNN_QUERY) */
    MM_STAGE_LOCATE_PASS0 = 6, /* the first locate pass alone (inside MM_STAGE_LOCATE) */
    MM_STAGE_COUNT = 7
};
/* on: 0 no stage timers; 1 all stages; 2 only MM_STAGE_KNN_CELL and MM_STAGE_LOCATE_PASS0 (the two dominant
 * kernels: every timed stage costs the stream two events, ~5 us each between kernels) */
int mm_set_profiling(mm_context *ctx, int on);
int mm_last_timings(mm_context *ctx, double *ms, int n);

#ifdef __cplusplus
}
#endif
#endif /* MULTIMESH_HIP_H */
