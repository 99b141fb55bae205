// Context, error reporting, scratch pool, stage timers and resident-array helpers.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "mm_common.h"

static thread_local char g_err[512] = "";
static thread_local int g_status = MM_OK;

void mm_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_status = code;
}

extern "C" const char *mm_last_error(void) { return g_err; }
extern "C" int mm_last_status(void) { return g_status; }
void mm_clear_status(void) { g_status = MM_OK; }

// ---- allocation (with the guarded variant of test runs) ------------------------------------------------
namespace {
struct GuardBlock {
    void *user, *va;
    size_t reserved, mapped;
    hipMemGenericAllocationHandle_t handle;
};
std::mutex g_guard_mutex;
std::vector<GuardBlock> g_guard_blocks;
}  // namespace

bool mm_guard_alloc(void)
{
    static const bool on = getenv("MM_GUARD_ALLOC") != nullptr && atoi(getenv("MM_GUARD_ALLOC")) != 0;
    return on;
}

hipError_t mm_raw_alloc(int device, void **out, size_t bytes)
{
    if (!mm_guard_alloc()) return hipMalloc(out, bytes);
    *out = nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = (size_t)2 << 20;
    // the array ends at the end of the mapping (16-byte granules: vector loads stay aligned); one granule
    // of reserved, unmapped addresses follows it
    GuardBlock b = {};
    const size_t user = (bytes + 15) & ~(size_t)15;
    b.mapped = (user + gran - 1) / gran * gran;
    b.reserved = b.mapped + gran;
    e = hipMemAddressReserve(&b.va, b.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&b.handle, b.mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(b.va, b.reserved); return e; }
    e = hipMemMap(b.va, b.mapped, 0, b.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(b.va, b.mapped, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(b.va, b.mapped);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(b.handle);
        (void)hipMemAddressFree(b.va, b.reserved);
        return e;
    }
    b.user = (char *)b.va + (b.mapped - user);
    {
        std::lock_guard<std::mutex> lock(g_guard_mutex);
        g_guard_blocks.push_back(b);
    }
    *out = b.user;
    return hipSuccess;
}

hipError_t mm_raw_free(void *ptr)
{
    if (!ptr) return hipSuccess;
    if (!mm_guard_alloc()) return hipFree(ptr);
    GuardBlock b = {};
    {
        std::lock_guard<std::mutex> lock(g_guard_mutex);
        size_t at = 0;
        while (at < g_guard_blocks.size() && g_guard_blocks[at].user != ptr) ++at;
        if (at == g_guard_blocks.size()) return hipErrorInvalidValue;
        b = g_guard_blocks[at];
        g_guard_blocks.erase(g_guard_blocks.begin() + (long)at);
    }
    // The address range is NOT given back: a later reservation would get the same addresses, and kernels were
    // seen to read through the stale translations of the previous mapping (zeros, or a fault at the old guard
    // page).  A test process spends a few hundred GB of its 128 TB of addresses instead; the memory itself is
    // released.
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemUnmap(b.va, b.mapped);
    if (e == hipSuccess) e = hipMemRelease(b.handle);
    return e;
}

static void scratch_release(mm_context *ctx)
{
    for (int q = 0; q < ctx->scratch.guard_pieces; ++q) (void)mm_raw_free(ctx->scratch.guard_piece[q]);
    ctx->scratch.guard_pieces = 0;
    if (ctx->scratch.base) (void)mm_raw_free(ctx->scratch.base);
    ctx->scratch.base = nullptr;
    ctx->scratch.capacity = 0;
}

extern "C" int mm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int mm_context_create(int device, void *hip_stream, mm_context **out)
{
    MM_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mm_set_error(MM_ERR_NODEVICE, "no usable GPU (hipGetDeviceCount found none); "
                                      "multi_mesh_hip has no CPU fallback");
        return MM_ERR_NODEVICE;
    }
    MM_REQUIRE(device >= 0 && device < n, "device index out of range");
    MM_HIP_CHECK(hipSetDevice(device));
    mm_context *ctx = new (std::nothrow) mm_context();
    if (!ctx) {
        mm_set_error(MM_ERR_ALLOC, "out of host memory");
        return MM_ERR_ALLOC;
    }
    ctx->device = device;
    ctx->stream = (hipStream_t)hip_stream;
    {
        const char *fp = getenv("MM_FP_MODE");
        if (fp && (fp[0] == 't' || fp[0] == 'T' || fp[0] == '1')) ctx->fp_mode = MM_FP_TOL;
    }
    hipError_t e = mm_raw_alloc(device, (void **)&ctx->d_counters, 64 * sizeof(i64));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, 64 * sizeof(i64));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_counters, 64 * sizeof(i64), 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_misc, hipEventDisableTiming);
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_HIP, "context allocation failed: %s", hipGetErrorString(e));
        if (ctx->d_counters) (void)mm_raw_free(ctx->d_counters);
        if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
        delete ctx;
        return MM_ERR_HIP;
    }
    for (int s = 0; s < MM_STAGE_COUNT; ++s) ctx->ev_used[s] = false;
    *out = ctx;
    return MM_OK;
}

extern "C" void mm_context_destroy(mm_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    scratch_release(ctx);
    for (int s = 0; s < MM_BUF_COUNT; ++s)
        if (ctx->buf_ptr[s]) (void)mm_raw_free(ctx->buf_ptr[s]);
    if (ctx->d_counters) (void)mm_raw_free(ctx->d_counters);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->ev_misc) (void)hipEventDestroy(ctx->ev_misc);
    for (int q = 0; q < 3; ++q)
        if (ctx->ev_copy[q]) (void)hipEventDestroy(ctx->ev_copy[q]);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->ev_created)
        for (int s = 0; s < MM_STAGE_COUNT; ++s) {
            (void)hipEventDestroy(ctx->ev_begin[s]);
            (void)hipEventDestroy(ctx->ev_end[s]);
        }
    delete ctx;
}

extern "C" int mm_synchronize(mm_context *ctx)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MM_OK;
}

extern "C" int mm_device_alloc(mm_context *ctx, size_t bytes, void **dptr)
{
    MM_REQUIRE(ctx != nullptr && dptr != nullptr, "null argument");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    *dptr = nullptr;
    if (bytes == 0) bytes = 256;
    hipError_t e = mm_raw_alloc(ctx->device, dptr, bytes);
    if (e != hipSuccess) {
        mm_set_error(MM_ERR_ALLOC, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return MM_ERR_ALLOC;
    }
    return MM_OK;
}

extern "C" int mm_device_free(mm_context *ctx, void *dptr)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    if (!dptr) return MM_OK;
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MM_HIP_CHECK(mm_raw_free(dptr));
    return MM_OK;
}

extern "C" int mm_copy_h2d(mm_context *ctx, void *dst_d, const void *src_h, size_t bytes)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    if (bytes == 0) return MM_OK;
    MM_REQUIRE(dst_d != nullptr && src_h != nullptr, "null pointer");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MM_OK;
}

extern "C" int mm_copy_d2h(mm_context *ctx, void *dst_h, const void *src_d, size_t bytes)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    if (bytes == 0) return MM_OK;
    MM_REQUIRE(dst_h != nullptr && src_d != nullptr, "null pointer");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MM_OK;
}

extern "C" int mm_memset(mm_context *ctx, void *dst_d, int value, size_t bytes)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    if (bytes == 0) return MM_OK;
    MM_REQUIRE(dst_d != nullptr, "null pointer");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    MM_HIP_CHECK(hipMemsetAsync(dst_d, value, bytes, ctx->stream));
    return MM_OK;
}

// ---- clearing device memory on the hot path -------------------------------------------
// hipMemsetAsync is a dispatch like any other (a fill kernel of the runtime's), but the runtime queues it behind a
// barrier of its own: in the per-dispatch timeline of a step every fill came 11 us after the kernel before it, three
// times per step.  A kernel of ours is just the next packet in the queue.
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned char *__restrict__ dst, size_t bytes)
{
    // 16 bytes per thread and trip where the span allows (dst is 16-byte aligned: scratch carves and the counter
    // block are 256-byte aligned), single bytes for an odd tail (guarded allocations end with their array)
    const size_t n16 = bytes / 16;
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += (size_t)gridDim.x * blockDim.x)
        d4[q] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[n16 * 16 + threadIdx.x] = 0;
}

int mm_zero_async(mm_context *ctx, void *dst_d, size_t bytes)
{
    if (bytes == 0) return MM_OK;
    if ((reinterpret_cast<uintptr_t>(dst_d) & 15) != 0) {   // (never on the hot path)
        MM_HIP_CHECK(hipMemsetAsync(dst_d, 0, bytes, ctx->stream));
        return MM_OK;
    }
    const size_t n16 = bytes / 16;
    const size_t want = (n16 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<unsigned char *>(dst_d), bytes);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

// n 64-bit words from device memory to the context's pinned host mirror, by a kernel (a copy command of the runtime's
// is a dispatch of 4 us behind a 6 us gap; this is 4 us and no gap).  The host reads them after synchronising.
__global__ void mirror_words_kernel(const long long *__restrict__ src, long long *__restrict__ dst, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

int mm_mirror_async(mm_context *ctx, long long *dst_pinned, const long long *src_d, int n)
{
    hipLaunchKernelGGL(mirror_words_kernel, dim3(1), dim3(64), 0, ctx->stream, src_d, dst_pinned, n < 64 ? n : 64);
    MM_HIP_CHECK(hipGetLastError());
    return MM_OK;
}

// ---- scratch -----------------------------------------------------------------------
int mm_scratch_begin(mm_context *ctx, size_t total)
{
    total = mm_round256(total) + 4096;
    if (mm_guard_alloc()) {
        // every carve is an allocation of its own, given back when the next call begins
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        scratch_release(ctx);
        ctx->scratch.capacity = total;   // (the carves are still checked against the reservation)
        ctx->scratch.used = 0;
        return MM_OK;
    }
    if (total > ctx->scratch.capacity) {
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch.base) MM_HIP_CHECK(hipFree(ctx->scratch.base));
        ctx->scratch.base = nullptr;
        ctx->scratch.capacity = 0;
        size_t want = total + total / 8;
        hipError_t e = hipMalloc((void **)&ctx->scratch.base, want);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_ALLOC, "scratch hipMalloc(%zu) failed: %s", want,
                         hipGetErrorString(e));
            return MM_ERR_ALLOC;
        }
        ctx->scratch.capacity = want;
    }
    ctx->scratch.used = 0;
    return MM_OK;
}

void *mm_scratch_take(mm_context *ctx, size_t bytes)
{
    if (mm_guard_alloc()) {
        if (ctx->scratch.used + mm_round256(bytes) > ctx->scratch.capacity) return nullptr;
        if (ctx->scratch.guard_pieces >= (int)(sizeof(ctx->scratch.guard_piece) / sizeof(ctx->scratch.guard_piece[0]))) {
            fprintf(stderr, "multi_mesh_hip: MM_GUARD_ALLOC: more scratch carves in one call than the guard tracks (not a pool exhaustion)\n");
            return nullptr;
        }
        void *p = nullptr;
        if (mm_raw_alloc(ctx->device, &p, bytes > 0 ? bytes : 16) != hipSuccess) return nullptr;
        ctx->scratch.guard_piece[ctx->scratch.guard_pieces++] = p;
        ctx->scratch.used += mm_round256(bytes);
        return p;
    }
    bytes = mm_round256(bytes);
    if (ctx->scratch.used + bytes > ctx->scratch.capacity) return nullptr;
    void *p = ctx->scratch.base + ctx->scratch.used;
    ctx->scratch.used += bytes;
    return p;
}

int mm_buffer_get(mm_context *ctx, int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 256;
    // (guarded runs: exactly the size asked for, so that the array ends with its mapping)
    if (bytes > ctx->buf_cap[slot] || (mm_guard_alloc() && bytes != ctx->buf_cap[slot])) {
        MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->buf_ptr[slot]) MM_HIP_CHECK(mm_raw_free(ctx->buf_ptr[slot]));
        ctx->buf_ptr[slot] = nullptr;
        ctx->buf_cap[slot] = 0;
        const size_t want = mm_guard_alloc() ? bytes : mm_round256(bytes + bytes / 16);
        hipError_t e = mm_raw_alloc(ctx->device, &ctx->buf_ptr[slot], want);
        if (e != hipSuccess) {
            mm_set_error(MM_ERR_ALLOC, "hipMalloc(%zu) for pipeline buffer %d failed: %s", want, slot,
                         hipGetErrorString(e));
            return MM_ERR_ALLOC;
        }
        ctx->buf_cap[slot] = want;
    }
    *out = ctx->buf_ptr[slot];
    return MM_OK;
}

// ---- stage timers ------------------------------------------------------------------
extern "C" int mm_set_profiling(mm_context *ctx, int on)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    if (on && !ctx->ev_created) {
        for (int s = 0; s < MM_STAGE_COUNT; ++s) {
            MM_HIP_CHECK(hipEventCreate(&ctx->ev_begin[s]));
            MM_HIP_CHECK(hipEventCreate(&ctx->ev_end[s]));
        }
        ctx->ev_created = true;
    }
    ctx->profiling = on == 2 ? 2 : (on ? 1 : 0);   // 2: the two dominant kernels only (see the header)
    return MM_OK;
}

extern "C" int mm_set_lazy_lists(mm_context *ctx, int on)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    ctx->lazy_lists = on ? 1 : 0;
    return MM_OK;
}

extern "C" int mm_set_fp_mode(mm_context *ctx, int mode)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    MM_REQUIRE(mode == MM_FP_EXACT || mode == MM_FP_TOL, "mode must be MM_FP_EXACT or MM_FP_TOL");
    ctx->fp_mode = mode;
    return MM_OK;
}

extern "C" int mm_get_fp_mode(mm_context *ctx)
{
    MM_REQUIRE(ctx != nullptr, "ctx is null");
    return ctx->fp_mode;
}

// {solves of the last hex8 locate stage that MM_FP_TOL could not certify and repeated in the reference's arithmetic,
//  targets that went through the reference-order kernel (fallback / failure candidates), 0, 0}.  Synchronises.
extern "C" int mm_last_locate_stats(mm_context *ctx, long long *out4)
{
    MM_REQUIRE(ctx != nullptr && out4 != nullptr, "null argument");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    int h[16];
    MM_HIP_CHECK(hipMemcpyAsync(h, reinterpret_cast<const int *>(ctx->d_counters + 8), sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    out4[0] = h[13];
    out4[1] = h[15];
    out4[2] = h[14];
    out4[3] = 0;
    return MM_OK;
}

void mm_stage_reset(mm_context *ctx)
{
    for (int s = 0; s < MM_STAGE_COUNT; ++s) ctx->ev_used[s] = false;
}

// (an event between two kernels costs the stream ~5 us: a step with all seven stages timed is ~40 us longer)
static inline bool stage_timed(const mm_context *ctx, int stage)
{
    return ctx->profiling == 1 || (ctx->profiling == 2 && (stage == MM_STAGE_KNN_CELL || stage == MM_STAGE_LOCATE_PASS0));
}

void mm_stage_begin(mm_context *ctx, int stage)
{
    if (!stage_timed(ctx, stage)) return;
    (void)hipEventRecord(ctx->ev_begin[stage], ctx->stream);
}

void mm_stage_end(mm_context *ctx, int stage)
{
    if (!stage_timed(ctx, stage)) return;
    (void)hipEventRecord(ctx->ev_end[stage], ctx->stream);
    ctx->ev_used[stage] = true;
}

extern "C" int mm_last_timings(mm_context *ctx, double *ms, int n)
{
    MM_REQUIRE(ctx != nullptr && ms != nullptr, "null argument");
    MM_HIP_CHECK(hipSetDevice(ctx->device));
    for (int s = 0; s < n; ++s) ms[s] = 0.0;
    if (!ctx->profiling) return MM_STAGE_COUNT;
    MM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < MM_STAGE_COUNT && s < n; ++s) {
        if (!ctx->ev_used[s]) continue;
        float t = 0.f;
        MM_HIP_CHECK(hipEventElapsedTime(&t, ctx->ev_begin[s], ctx->ev_end[s]));
        ms[s] = (double)t;
    }
    return MM_STAGE_COUNT;
}
