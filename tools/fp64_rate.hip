// Microbenchmark: issue rate of plain fp64 add / mul / fma on gfx950 (no contraction).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; x4 = x4 + a; x5 = x5 + a; x6 = x6 + a; x7 = x7 + a; }
        if (MODE == 1) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a; }
        if (MODE == 2) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
                         x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b); }
        if (MODE == 3) { float f0 = x0, f1 = x1; f0 = f0 + (float)a; f1 = f1 * (float)a; x0 = f0; x1 = f1;  }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    double *d; hipMalloc(&d, 256 * 2048 * 8 * 8);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) for (int blocks : {256, 1024, 2048, 4096}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 0.5);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 0.5);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 0.5);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winst = (double)blocks * 4 * iters * 8;              // wave-instructions
        double per_simd = winst / 1024.0;                            // per SIMD
        printf("mode %d blocks %d: %.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz (waves/SIMD=%.1f)\n", mode, blocks, ms,
               ms * 1e-3 * 2.4e9 / per_simd, blocks * 4 / 1024.0);
    }
    return 0;
}
