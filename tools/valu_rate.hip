// Microbenchmark: issue cost of the fp32 VALU instructions the kNN lane kernel is made of, on gfx950:
// v_fma_f32, v_pk_fma_f32 (two lanes' worth per instruction), v_med3_f32, v_and_or_b32 -- cycles per
// wave-instruction per SIMD at 1, 2, 4 and 8 waves per SIMD (8 independent chains per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8];
    v2f y[8];
    for (int u = 0; u < 8; ++u) { x[u] = threadIdx.x + u; y[u] = v2f{x[u], x[u] + 0.5f}; }
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) x[u] = __builtin_fmaf(x[u], a, b);
            if (MODE == 1) y[u] = __builtin_elementwise_fma(y[u], a2, b2);
            if (MODE == 2) x[u] = __builtin_amdgcn_fmed3f(x[u], a, b);
            if (MODE == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(a), "v"(b));
            if (MODE == 4) y[u] = y[u] * a2;
            if (MODE == 5) y[u] = y[u] + a2;
        }
        if (MODE == 6) {   // compare-exchange pairs: v_min_f32 + v_max_f32, both operands in VGPRs
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                float lo, hi;
                asm volatile("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(x[u]), "v"(x[u + 1]));
                x[u] = lo;
                x[u + 1] = hi;
            }
        }
        if (MODE == 7) {   // v_med3_f32, three VGPR operands
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = __builtin_amdgcn_fmed3f(x[u], x[(u + 1) & 7], x[(u + 2) & 7]);
        }
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += x[u] + y[u].x + y[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8192 * 4);
    const int iters = 20000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_med3_f32", "v_and_or_b32", "v_pk_mul_f32", "v_pk_add_f32",
                           "v_min+v_max", "v_med3 3vgpr"};
    for (int mode = 0; mode < 8; ++mode) for (int blocks : {256, 512, 1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;   // wave-instructions per SIMD
        printf("%-13s %.1f waves/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (= cycles / GHz)\n", names[mode],
               blocks * 4 / 1024.0, ms, ms * 1e6 / per_simd);
    }
    return 0;
}
