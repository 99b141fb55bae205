// A2, the density-adaptive index (round 4): a linear octree over MORTON-SORTED sources, for clouds whose density varies
// by orders of magnitude (graded meshes).  A part of mm_knn.hip -- ONE translation unit --, included inside that file's
// anonymous namespace after the tiled kernels and before the lane kernel, whose TREE instantiation walks it.
//
// Why.  The stack of uniform grids (mm_knn_build_impl, "density levels") serves a target only where a level's design
// density is within ~0.5x .. 1.2x of the local one, every level is a dense array over the whole box, and a target falls
// through the levels one counting sort at a time (graded 10M mesh, u -> u^1.5: 12.3 ms per pass against 3.4 uniform).
// Here there are no levels to lay out: the sources are sorted ONCE by the Morton code of their position (16 bits per
// axis over the bounding cube), and then EVERY octree cell at EVERY level is one contiguous run of that array, found by
// two binary searches on the keys (started from a table of the level-7 cells).  What plays the part of a grid cell is
// chosen per place:
//   leaf    : for source i, the shortest key prefix (a BINARY node: an octree cell, or one cut in half along x, or along x
//             and y) whose node holds at most kTreeNmax sources (tree_leaf_level_kernel: a node of P bits holds more than N
//             sources iff some pair of keys N apart in the sorted order share P bits -- a sliding maximum of common-prefix
//             lengths, no tree is ever linked).
//   item    : the targets, sorted by the same code, fall into runs that share a node -- chosen per target among the leaf
//             around it (or, where the tree has an EMPTY child: hulls, strongly anisotropic lattices, the largest empty node)
//             and its three nearest ancestors: the one with the most sources whose window certifies on paper and fits the
//             tile (tree_target_node_kernel, see the table of window shapes below); a run of at most 256 targets of one node
//             is a work item of the lane kernel.
//   window  : the node plus one cubic cell of margin on every side; one lane per cell (up to four cells per lane) does two
//             searches on the keys; the runs are staged into the LDS tile exactly as the uniform kernel stages its cells
//             (thin layers along z and all), and the scan / exact distances / certificate are that kernel's: a row is
//             accepted only if its exact k-th distance is below the distance to every face of the window that has cells
//             behind it.
//   passes  : what the first windows do not serve (too full for the tile, a k-th neighbour beyond the margin) is sorted
//             again and served by windows with wider margins (tree_query, second_pass); what is left after that takes the
//             single-target search (tree_ring_kernel; short lists: one wave per target over the level-0 grid).
// Results do not depend on any of this (tests: test_knn_tree_* in tests/test_parity_gpu.py, tools/fuzz_knn.py with
// MM_KNN_TREE=1 forcing the tree on every cloud).  Where it stands: mm_knn.hip, tree_mode.
typedef unsigned long long u64;

constexpr int kTreeQ = 16;                            // bits per axis: 48-bit keys
constexpr int kTreeBits = 3 * kTreeQ;
constexpr int kTreeL0 = 7;                            // the search table: first source of every level-7 cell (2^21 + 2 entries)
constexpr int kTreeShift0 = 3 * (kTreeQ - kTreeL0);
constexpr int kTreeCoarse = 1 << (3 * kTreeL0);
constexpr int kTreeNmax = 64;                         // sources per leaf
// Nodes are BINARY: a node is a prefix of P bits of the key (0 .. 48), i.e. the octree cell of level P / 3 cut in half
// along x (P % 3 >= 1) and along y (P % 3 == 2) -- counts come in steps of 2x, not 8x, so a node of the wanted size exists
// everywhere.  A window is the node plus ONE cell of margin on every side, in cubic cells of level P / 3 + d (x, y, z):
//   P % 3 = 0, d = 2 : node 4 x 4 x 4 cells, window 6 x 6 x 6 (3.4 x the node's sources)      d = 1 : 2 x 2 x 2 -> 4 x 4 x 4 (8 x)
//   P % 3 = 1        :      2 x 4 x 4,              4 x 6 x 6 (4.5 x)                                  1 x 2 x 2 -> 3 x 4 x 4 (12 x)
//   P % 3 = 2        :      2 x 2 x 4,              4 x 4 x 6 (6 x)                                    1 x 1 x 2 -> 3 x 3 x 4 (18 x)
//   P % 3 = 0, d = 0 : the node is one cell, window 3 x 3 x 3 (27 x)
// The margin certifies a list of k when a cell holds enough sources (its edge against the local spacing: ~2 for k <= 8,
// ~8 for k = 20), and the window must fit the tile.  tree_target_node_kernel takes, among the leaf around a target and its
// three nearest ancestors, the node with the most sources that has such a window (most targets per work item).
constexpr int kTreeLevelShift = 56;                   // a node = its first key | P << 56 | d << 62
constexpr int kTreeTileMax = 700;                     // expected sources of a window (the tile holds kLaneTileCap = 768)
__host__ __device__ __forceinline__ int tree_cell_min(int k) { return k <= 8 ? 2 : 8; }
// sources the window (P, d) of a node with n sources is expected to hold; -1: no such window
__device__ __forceinline__ int tree_window_load(int n, int P, int d)
{
    const int j = P % 3;
    if ((d == 0 && j != 0) || P / 3 + d > kTreeQ) return -1;
    const int a = 1 << d;
    const int ax = j >= 1 ? a >> 1 : a, ay = j >= 2 ? a >> 1 : a, az = a;
    return (int)(((i64)n * ((ax + 2) * (ay + 2) * (az + 2))) / (ax * ay * az));
}
__device__ __forceinline__ int tree_node_cells(int P, int d) { return (1 << (3 * d)) >> (P % 3); }

struct TreeParams {
    double lox, loy, loz;
    double scale;   // 2^Q / edge of the bounding cube
    double size;    // edge of the bounding cube
};

struct TreeItem {
    u64 node;       // first key | prefix length << 56 | cell depth d << 62
    int t0;         // first target (position in the Morton-sorted targets)
    int nsrc;       // sources in the node
};

struct TreeArgs {
    TreeParams tp;
    const u64 *keys;        // [nsrc] sorted
    const int *coarse;      // [kTreeCoarse + 1]
    const TreeItem *items;  // [nitems + 1] (the last one only carries t0 = number of targets)
    const int *nitems;      // device
};

__device__ __forceinline__ u64 tree_spread3(unsigned v)
{
    u64 x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ unsigned tree_compact3(u64 x)
{
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return (unsigned)x;
}

// x takes the most significant bit of every triple, z the least: a binary node is cut in half along x first, then y --
// what is left of the cell is a COLUMN along z, the axis the lane kernel's thin layers trim a target's window along
// (with z cut first the half-cells were slabs across z: 40 % more candidates per target on the graded 10M mesh)
__device__ __forceinline__ u64 tree_morton(unsigned x, unsigned y, unsigned z)
{
    return tree_spread3(z) | (tree_spread3(y) << 1) | (tree_spread3(x) << 2);
}
__device__ __forceinline__ unsigned tree_key_x(u64 k) { return tree_compact3(k >> 2); }
__device__ __forceinline__ unsigned tree_key_y(u64 k) { return tree_compact3(k >> 1); }
__device__ __forceinline__ unsigned tree_key_z(u64 k) { return tree_compact3(k); }

// the finest cell of a coordinate: monotone in v (sources and targets go through the same arithmetic); points beyond the
// cube -- targets outside the sources' box -- sit in the boundary cells, NaN in cell 0
__device__ __forceinline__ unsigned tree_quant(double v, double lo, double scale)
{
    const double f = (v - lo) * scale;
    const double top = (double)((1 << kTreeQ) - 1);
    return f >= top ? (unsigned)((1 << kTreeQ) - 1) : (f > 0.0 ? (unsigned)(int)f : 0u);
}

__device__ __forceinline__ u64 tree_key_of(double x, double y, double z, const TreeParams &tp)
{
    return tree_morton(tree_quant(x, tp.lox, tp.scale), tree_quant(y, tp.loy, tp.scale), tree_quant(z, tp.loz, tp.scale));
}

// leading key bits two keys share (0 .. 48)
__device__ __forceinline__ int tree_common_bits(u64 a, u64 b)
{
    const u64 x = a ^ b;
    return x == 0 ? kTreeBits : __clzll((long long)x) - (64 - kTreeBits);
}

__device__ __forceinline__ bool tree_same_node(u64 a, u64 b, int bits)
{
    return ((a ^ b) >> (kTreeBits - bits)) == 0;
}

// first source whose key is >= key
__device__ __forceinline__ int tree_lower_bound(const u64 *__restrict__ keys, const int *__restrict__ coarse, int nsrc, u64 key)
{
    if (key >> kTreeBits) return nsrc;
    const int c = (int)(key >> kTreeShift0);
    int lo = coarse[c], hi = coarse[c + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// sources of the node of `bits` leading bits around key
__device__ __forceinline__ int tree_node_count(const u64 *__restrict__ keys, const int *__restrict__ coarse, int nsrc, u64 key, int bits)
{
    const int sh = kTreeBits - bits;
    const u64 k0 = (key >> sh) << sh;
    return tree_lower_bound(keys, coarse, nsrc, k0 + (1ull << sh)) - tree_lower_bound(keys, coarse, nsrc, k0);
}

// the leaf around key (its prefix length), or the largest empty node around it: p = tree_lower_bound(key)
__device__ __forceinline__ int tree_leaf_bits(const u64 *__restrict__ keys, const unsigned char *__restrict__ sbits, int nsrc,
                                              u64 key, int p)
{
    const u64 ka = p < nsrc ? keys[p] : 0ull, kb = p > 0 ? keys[p - 1] : 0ull;
    if (p < nsrc && tree_same_node(ka, key, sbits[p])) return sbits[p];
    if (p > 0 && tree_same_node(kb, key, sbits[p - 1])) return sbits[p - 1];
    int m = -1;
    if (p < nsrc) m = max(m, tree_common_bits(ka, key));
    if (p > 0) m = max(m, tree_common_bits(kb, key));
    return min(m + 1, kTreeBits);
}

constexpr int kTreeScanMax = 32;   // runs up to this long are scanned whole by the single-target search

// squared distance from p to the box of the node (first key k0, P leading bits); 0 inside.  (slack: see the lane kernel)
__device__ __forceinline__ double tree_box_dist2(const TreeParams &tp, u64 k0, int P, double px, double py, double pz)
{
    const int l = P / 3, j = P - 3 * l;
    const double cell0 = tp.size / (double)(1 << kTreeQ);
    const double e = tp.size / (double)(1 << l);
    const double ex = j >= 1 ? 0.5 * e : e, ey = j >= 2 ? 0.5 * e : e, ez = e;
    const double xl = tp.lox + (double)tree_key_x(k0) * cell0, yl = tp.loy + (double)tree_key_y(k0) * cell0,
                 zl = tp.loz + (double)tree_key_z(k0) * cell0;
    const double slack = 1e-9 * e;
    const double ddx = fmax(fmax(xl - px, px - (xl + ex)) - slack, 0.0);
    const double ddy = fmax(fmax(yl - py, py - (yl + ey)) - slack, 0.0);
    const double ddz = fmax(fmax(zl - pz, pz - (zl + ez)) - slack, 0.0);
    return ddx * ddx + ddy * ddy + ddz * ddz;
}

// keys and row numbers of the points (list: of the listed ones; entries behind the list's end sort last)
__global__ __launch_bounds__(kBlock) void tree_keys_kernel(const double *__restrict__ pts, i64 n, int ndim, int pstride, TreeParams tp,
                                                           const int *__restrict__ list, const int *__restrict__ list_count,
                                                           u64 *__restrict__ keys, unsigned *__restrict__ vals)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (list && i >= (i64)*list_count) {
        keys[i] = ~0ull;
        vals[i] = 0u;
        return;
    }
    const i64 p = list ? (i64)list[i] : i;
    const double x = pts[p * pstride], y = ndim > 1 ? pts[p * pstride + 1] : 0.0, z = ndim > 2 ? pts[p * pstride + 2] : 0.0;
    keys[i] = tree_key_of(x, y, z, tp);
    vals[i] = (unsigned)p;
}

// the sorted order as 32-byte records {x, y, z, row}
__global__ __launch_bounds__(kBlock) void tree_records_kernel(const double *__restrict__ pts, int ndim, int pstride,
                                                              const unsigned *__restrict__ vals, i64 n,
                                                              const int *__restrict__ count, double *__restrict__ recs)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (count ? (i64)*count : n)) return;
    const i64 p = vals[i];
    const double x = pts[p * pstride], y = ndim > 1 ? pts[p * pstride + 1] : 0.0, z = ndim > 2 ? pts[p * pstride + 2] : 0.0;
    store_record(recs + i * kRec, x, y, z, (int)p);
}

// coarse[c] = first source of level-7 cell c or of any later one; coarse[kTreeCoarse] = nsrc
__global__ __launch_bounds__(kBlock) void tree_coarse_kernel(const u64 *__restrict__ keys, int nsrc, int *__restrict__ coarse)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > kTreeCoarse) return;
    const u64 key = (u64)c << kTreeShift0;
    int lo = 0, hi = nsrc;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    coarse[c] = lo;
}

// level[i] = prefix length of the leaf around source i: one more bit than the longest prefix at which the node around it
// still holds more than kTreeNmax sources -- i.e. at which some j in [i - N, i] has keys j and j + N in one node
__global__ __launch_bounds__(kBlock) void tree_leaf_level_kernel(const u64 *__restrict__ keys, i64 n,
                                                                 unsigned char *__restrict__ level)
{
    __shared__ u64 s_key[kBlock + 2 * kTreeNmax];
    __shared__ signed char s_c[kBlock + kTreeNmax];
    const i64 base = (i64)blockIdx.x * kBlock;
    for (int q = threadIdx.x; q < kBlock + 2 * kTreeNmax; q += kBlock) {
        const i64 j = base - kTreeNmax + q;
        s_key[q] = (j >= 0 && j < n) ? keys[j] : 0ull;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < kBlock + kTreeNmax; q += kBlock) {
        const i64 j = base - kTreeNmax + q;
        s_c[q] = (j >= 0 && j + kTreeNmax < n) ? (signed char)tree_common_bits(s_key[q], s_key[q + kTreeNmax]) : (signed char)-1;
    }
    __syncthreads();
    const i64 i = base + threadIdx.x;
    if (i >= n) return;
    int m = -1;
    for (int d = 0; d <= kTreeNmax; ++d) m = max(m, (int)s_c[threadIdx.x + d]);
    level[i] = (unsigned char)min(m + 1, kTreeBits);
}

// node[t] of sorted target t: the leaf its key lies in, or the largest empty node around it
__global__ __launch_bounds__(kBlock) void tree_target_node_kernel(const u64 *__restrict__ tkeys, i64 nt,
                                                                  const int *__restrict__ count,
                                                                  const u64 *__restrict__ skeys, int nsrc,
                                                                  const int *__restrict__ coarse,
                                                                  const unsigned char *__restrict__ slevel, int cell_min,
                                                                  int tile_max, u64 *__restrict__ node)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (count ? (i64)*count : nt)) return;
    const u64 key = tkeys[t];
    const int p = tree_lower_bound(skeys, coarse, nsrc, key);
    const int leaf = tree_leaf_bits(skeys, slevel, nsrc, key, p);
    // the leaf and its three nearest ancestors: the node with the most sources that has a certifying window in the tile
    int bestP = -1, bestd = 0, bestn = -1, n0 = 0;
    for (int up = 0; up <= 3 && leaf - up >= 0; ++up) {
        const int P = leaf - up;
        const int n = tree_node_count(skeys, coarse, nsrc, key, P);
        if (up == 0) n0 = n;
        if (n <= bestn) continue;
        for (int d = 2; d >= 0; --d) {
            const int load = tree_window_load(n, P, d);
            if (load < 0 || load > tile_max || n < cell_min * tree_node_cells(P, d)) continue;
            bestP = P;
            bestd = d;
            bestn = n;
            break;
        }
    }
    if (bestP < 0) {
        // nothing certifies on paper (a sparse or empty leaf beside dense ones): the leaf with its widest window that fits
        bestP = leaf;
        bestd = leaf % 3 != 0 ? 1 : 0;
        for (int d = bestd; d <= 2; ++d) {
            const int load = tree_window_load(n0, leaf, d);
            if (load >= 0 && load <= tile_max) {
                bestd = d;
                break;
            }
        }
        if (leaf / 3 + bestd > kTreeQ) {   // (the finest cells: no deeper ones to cut a window from)
            bestP = 3 * (leaf / 3);
            bestd = 0;
        }
    }
    const int sh = kTreeBits - bestP;
    node[t] = ((key >> sh) << sh) | ((u64)bestP << kTreeLevelShift) | ((u64)bestd << 62);
}

// flag[t] = 1 where a work item starts: a new node, or kTreePerItem targets into the array
__global__ __launch_bounds__(kBlock) void tree_item_flags_kernel(const u64 *__restrict__ node, i64 nt, const int *__restrict__ count,
                                                                 int per_item, int *__restrict__ flag)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const i64 n = count ? (i64)*count : nt;
    flag[t] = t < n && (t == 0 || node[t] != node[t - 1] || (t % per_item) == 0) ? 1 : 0;
}

// the items (rank = exclusive scan of the flags, rank[nt] = their number), each with its node's source count
__global__ __launch_bounds__(kBlock) void tree_items_kernel(const u64 *__restrict__ node, i64 nt, const int *__restrict__ count,
                                                            const int *__restrict__ flag, const int *__restrict__ rank,
                                                            const u64 *__restrict__ skeys, int nsrc,
                                                            const int *__restrict__ coarse, TreeItem *__restrict__ items,
                                                            int *__restrict__ nitems)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    if (t == 0) {
        TreeItem last;
        last.node = 0;
        last.t0 = (int)(count ? (i64)*count : nt);
        last.nsrc = 0;
        items[rank[nt]] = last;
        *nitems = rank[nt];
    }
    if (!flag[t]) return;
    const u64 nd = node[t];
    const int bits = (int)((nd >> kTreeLevelShift) & 0x3f);
    TreeItem it;
    it.node = nd;
    it.t0 = (int)t;
    it.nsrc = tree_node_count(skeys, coarse, nsrc, nd & ((1ull << kTreeLevelShift) - 1ull), bits);
    items[rank[t]] = it;
}

// The tree's search for single targets (what the windows of the lane kernel could not certify, long lists of them): one
// lane per target, the generic kernel's rule on the tree's cells -- the 3 x 3 x 3 block of octree cells around the target, its
// own cell first and every other one only if it can hold something nearer than the k-th so far, accepted when the k-th
// distance is below the block's faces; else the block of cells of twice the edge.  Measured on the way (graded 10M mesh,
// hand-overs of a k = 8 query, ns per target): blocks of leaf-sized cells found by searches 2.3 - 6.5 (u^2.2: 4.5, k = 20: 14),
// shells of same-size cells 4.7, a ball around the k nearest in Morton order cut into cells 24.
template <int K, typename IDX>
__global__ __launch_bounds__(kBlock) void tree_ring_kernel(TreeParams tp, const u64 *__restrict__ keys,
                                                           const int *__restrict__ coarse,
                                                           const unsigned char *__restrict__ slevel,
                                                           const double *__restrict__ xyz, int nsrc,
                                                           const double *__restrict__ pts, int ndim, int pstride, int kout,
                                                           IDX *__restrict__ idx_out, double *__restrict__ dist_out,
                                                           const int *__restrict__ list, const int *__restrict__ list_count,
                                                           int list_min)
{
    const i64 total = (i64)*list_count;
    if (total <= list_min) return;   // (short lists: one WAVE per target over the level-0 grid, see tree_ring)
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const i64 i = (i64)list[q];
        const double px = pts[i * pstride];
        const double py = ndim > 1 ? pts[i * pstride + 1] : 0.0;
        const double pz = ndim > 2 ? pts[i * pstride + 2] : 0.0;
        const unsigned qx = tree_quant(px, tp.lox, tp.scale), qy = tree_quant(py, tp.loy, tp.scale), qz = tree_quant(pz, tp.loz, tp.scale);
        const u64 key = tree_morton(qx, qy, qz);
        // Blocks of cells of level <= kTreeL0 only: their runs come straight out of the search table (no search on the keys),
        // and inside a cell the pruned descent below goes wherever the k-th ball reaches -- in a dense place the target's own
        // cell is entered nearest-part-first and the 26 around it are skipped by their boxes, in an anisotropic one the
        // descent follows the ball instead of climbing level after level.  Start where the own cell holds k sources.
        int lvl = kTreeL0;
        while (lvl > 0) {
            const int c = (int)(key >> (3 * (kTreeQ - lvl))), up = 3 * (kTreeL0 - lvl);
            if (coarse[(c + 1) << up] - coarse[c << up] >= kout) break;
            --lvl;
        }
        BestList<K> best;
        int st_a[kTreeBits + 2], st_b[kTreeBits + 2], st_p[kTreeBits + 2];
        u64 st_k[kTreeBits + 2];
        for (int l = lvl;; --l) {
            best.init(nsrc);
            const int sh = kTreeQ - l;
            const int cx = (int)(qx >> sh), cy = (int)(qy >> sh), cz = (int)(qz >> sh);
            const int ncl = 1 << l;
            const double cl = tp.size / (double)ncl, slack = 1e-9 * cl;
            for (int ring = 0; ring <= 3; ++ring)
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (abs(dx) + abs(dy) + abs(dz) != ring) continue;   // (the cell itself, then faces, edges, corners)
                            const int gx = cx + dx, gy = cy + dy, gz = cz + dz;
                            if ((unsigned)gx >= (unsigned)ncl || (unsigned)gy >= (unsigned)ncl || (unsigned)gz >= (unsigned)ncl) continue;
                            if (ring > 0) {
                                double kth_now = best.d[K - 1];
                                if (kout < K) {
#pragma unroll
                                    for (int s = 0; s < K - 1; ++s)
                                        if (s == kout - 1) kth_now = best.d[s];
                                }
                                const double xl = tp.lox + (double)gx * cl, yl = tp.loy + (double)gy * cl, zl = tp.loz + (double)gz * cl;
                                const double ddx = fmax(fmax(xl - px, px - (xl + cl)) - slack, 0.0);
                                const double ddy = fmax(fmax(yl - py, py - (yl + cl)) - slack, 0.0);
                                const double ddz = fmax(fmax(zl - pz, pz - (zl + cl)) - slack, 0.0);
                                if (ddx * ddx + ddy * ddy + ddz * ddz > kth_now) continue;
                            }
                            const u64 cm = tree_morton((unsigned)gx, (unsigned)gy, (unsigned)gz);
                            const u64 k0 = cm << (3 * sh);
                            const int s0 = coarse[(int)cm << (3 * (kTreeL0 - l))], s1 = coarse[((int)cm + 1) << (3 * (kTreeL0 - l))];
                            // The cell's sources, nearest parts first: a run of more than kTreeScanMax sources is cut in two
                            // at the next key bit (one search inside the run) and a half is only entered if its box can hold
                            // something nearer than the k-th so far -- beside a much denser region a cell of the target's own
                            // size can hold 10^5 sources, of which a handful matter.
                            int top = 0;
                            st_a[0] = s0;
                            st_b[0] = s1;
                            st_k[0] = k0;
                            st_p[0] = 3 * l;
                            top = s1 > s0 ? 1 : 0;
                            while (top > 0) {
                                --top;
                                const int a = st_a[top], b = st_b[top], P = st_p[top];
                                const u64 kk = st_k[top];
                                {
                                    // (entered only if still worth it: the k-th distance may have shrunk since the push)
                                    double kth_now = best.d[K - 1];
                                    if (kout < K) {
#pragma unroll
                                        for (int s = 0; s < K - 1; ++s)
                                            if (s == kout - 1) kth_now = best.d[s];
                                    }
                                    if (tree_box_dist2(tp, kk, P, px, py, pz) > kth_now) continue;
                                }
                                if (b - a <= kTreeScanMax || P >= kTreeBits) {
                                    for (int s = a; s < b; s += 4) {
                                        double2 xy[4], zw[4];
#pragma unroll
                                        for (int u = 0; u < 4; ++u) {
                                            const double2 *r2 = reinterpret_cast<const double2 *>(xyz + (i64)min(s + u, b - 1) * kRec);
                                            xy[u] = r2[0];
                                            zw[u] = r2[1];
                                        }
#pragma unroll
                                        for (int u = 0; u < 4; ++u) {
                                            if (s + u < b) {
                                                const double ex = xy[u].x - px;
                                                const double ey = xy[u].y - py;
                                                const double ez = zw[u].x - pz;
                                                double d2 = ex * ex;
                                                d2 = d2 + ey * ey;
                                                if (ndim > 2) d2 = d2 + ez * ez;
                                                const int sid = record_id(zw[u].y);
                                                if (before(d2, sid, best.d[K - 1], best.id[K - 1])) best.insert(d2, sid);
                                            }
                                        }
                                    }
                                    continue;
                                }
                                const u64 kmid = kk | (1ull << (kTreeBits - 1 - P));
                                int lo = a, hi = b;
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (keys[mid] < kmid) lo = mid + 1;
                                    else hi = mid;
                                }
                                // the half nearer to the target on top of the stack
                                const double dlo = tree_box_dist2(tp, kk, P + 1, px, py, pz), dhi = tree_box_dist2(tp, kmid, P + 1, px, py, pz);
                                const bool lo_first = dlo <= dhi;
                                if (lo_first ? b > lo : lo > a) {
                                    st_a[top] = lo_first ? lo : a;
                                    st_b[top] = lo_first ? b : lo;
                                    st_k[top] = lo_first ? kmid : kk;
                                    st_p[top] = P + 1;
                                    ++top;
                                }
                                if (lo_first ? lo > a : b > lo) {
                                    st_a[top] = lo_first ? a : lo;
                                    st_b[top] = lo_first ? lo : b;
                                    st_k[top] = lo_first ? kk : kmid;
                                    st_p[top] = P + 1;
                                    ++top;
                                }
                            }
                        }
            if (l == 0) break;
            // faces of the block that have cells of the cube behind them
            double bound = INFINITY;
            if (cx - 1 > 0) bound = fmin(bound, (px - (tp.lox + (double)(cx - 1) * cl)) - slack);
            if (cx + 1 < ncl - 1) bound = fmin(bound, ((tp.lox + (double)(cx + 2) * cl) - px) - slack);
            if (cy - 1 > 0) bound = fmin(bound, (py - (tp.loy + (double)(cy - 1) * cl)) - slack);
            if (cy + 1 < ncl - 1) bound = fmin(bound, ((tp.loy + (double)(cy + 2) * cl) - py) - slack);
            if (cz - 1 > 0) bound = fmin(bound, (pz - (tp.loz + (double)(cz - 1) * cl)) - slack);
            if (cz + 1 < ncl - 1) bound = fmin(bound, ((tp.loz + (double)(cz + 2) * cl) - pz) - slack);
            double kth = best.d[K - 1];
            if (kout < K) {
#pragma unroll
                for (int s = 0; s < K - 1; ++s)
                    if (s == kout - 1) kth = best.d[s];
            }
            if (!(bound < INFINITY) || (bound > 0.0 && kth < bound * bound)) break;
        }
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (s < kout) {
                idx_out[i * kout + s] = (IDX)best.id[s];
                if (dist_out) dist_out[i * kout + s] = sqrt(best.d[s]);
            }
        }
    }
}
