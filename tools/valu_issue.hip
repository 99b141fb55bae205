// Microbenchmark (round 3): what does a VALU wave-instruction cost on a gfx950 SIMD, in CYCLES, as a function of
// how many waves share the SIMD and of the opcode?  tools/valu_rate.hip measured ns only and let the compiler pick
// the registers; here every operand is a fixed physical register (so register banks can be ruled in or out) and the
// in-kernel clock is measured beside it (s_memtime ticks per s_memrealtime tick of 10 ns), so the result reads as
// cycles per wave-instruction per SIMD.  Eight independent instructions per loop trip.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue tools/valu_issue.hip && tools/valu_issue
//
// Findings (MI355X, profiles/r03_valu_issue.txt): at >= 3 waves per SIMD the VOP2 arithmetic (v_add/sub/mul/fmac_f32)
// issues every ~2.4-2.6 cycles, a three-VGPR v_fma_f32 every ~3.2, but v_min/v_max/v_med3/v_max3_f32, v_and_or_b32 and
// the packed fp32 forms every ~4.4-4.8: the sorted-insert chain of the kNN lane kernel is made of HALF-RATE
// instructions.  Register banks (number mod 4) make no difference.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CLOB                                                                                                              \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", \
        "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",  \
        "v43", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"

// two-source form: op vD, vD, vB (B in the next bank)
#define T2(op)                                                                                                  \
    op " v8, v8, v21\n" op " v9, v9, v22\n" op " v10, v10, v23\n" op " v11, v11, v24\n" op " v12, v12, v25\n" op \
       " v13, v13, v26\n" op " v14, v14, v27\n" op " v15, v15, v28\n"
// same, both sources in ONE bank
#define T2B(op)                                                                                                 \
    op " v8, v8, v20\n" op " v9, v9, v21\n" op " v10, v10, v22\n" op " v11, v11, v23\n" op " v12, v12, v24\n" op \
       " v13, v13, v25\n" op " v14, v14, v26\n" op " v15, v15, v27\n"
// three-source form: op vD, vA, vB, vD (three banks)
#define T3(op)                                                                                                           \
    op " v8, v21, v22, v8\n" op " v9, v22, v23, v9\n" op " v10, v23, v24, v10\n" op " v11, v24, v25, v11\n" op            \
       " v12, v25, v26, v12\n" op " v13, v26, v27, v13\n" op " v14, v27, v28, v14\n" op " v15, v28, v29, v15\n"
// three-source form, all in one bank
#define T3B(op)                                                                                                          \
    op " v8, v20, v24, v8\n" op " v9, v21, v25, v9\n" op " v10, v22, v26, v10\n" op " v11, v23, v27, v11\n" op            \
       " v12, v24, v28, v12\n" op " v13, v25, v29, v13\n" op " v14, v26, v30, v14\n" op " v15, v27, v31, v15\n"
// the insert chain of the lane kernel: d[s] = op(d[s-1], c, d[s]) over consecutive registers, c = v20
#define TCHAIN(op)                                                                                                       \
    op " v17, v16, v20, v17\n" op " v16, v15, v20, v16\n" op " v15, v14, v20, v15\n" op " v14, v13, v20, v14\n" op        \
       " v13, v12, v20, v13\n" op " v12, v11, v20, v12\n" op " v11, v10, v20, v11\n" op " v10, v9, v20, v10\n"
// compares into 8 different SGPR pairs
#define TCMP(op)                                                                                                         \
    op " s[40:41], v8, v21\n" op " s[42:43], v9, v22\n" op " s[44:45], v10, v23\n" op " s[46:47], v11, v24\n" op          \
       " s[48:49], v12, v25\n" op " s[50:51], v13, v26\n" op " s[52:53], v14, v27\n" op " s[54:55], v15, v28\n"
// conditional moves on 8 different SGPR pairs
#define TCND                                                                                                             \
    "v_cndmask_b32 v8, v8, v21, s[40:41]\n v_cndmask_b32 v9, v9, v22, s[42:43]\n v_cndmask_b32 v10, v10, v23, s[44:45]\n"  \
    "v_cndmask_b32 v11, v11, v24, s[46:47]\n v_cndmask_b32 v12, v12, v25, s[48:49]\n v_cndmask_b32 v13, v13, v26, s[50:51]\n" \
    "v_cndmask_b32 v14, v14, v27, s[52:53]\n v_cndmask_b32 v15, v15, v28, s[54:55]\n"
// 64-bit two-source form on register pairs
#define T2D(op)                                                                                                          \
    op " v[8:9], v[8:9], v[22:23]\n" op " v[10:11], v[10:11], v[24:25]\n" op " v[12:13], v[12:13], v[26:27]\n" op         \
       " v[14:15], v[14:15], v[28:29]\n" op " v[16:17], v[16:17], v[30:31]\n" op " v[18:19], v[18:19], v[32:33]\n" op     \
       " v[34:35], v[34:35], v[38:39]\n" op " v[36:37], v[36:37], v[40:41]\n"
#define T3D(op)                                                                                                          \
    op " v[8:9], v[22:23], v[24:25], v[8:9]\n" op " v[10:11], v[24:25], v[26:27], v[10:11]\n" op                          \
       " v[12:13], v[26:27], v[28:29], v[12:13]\n" op " v[14:15], v[28:29], v[30:31], v[14:15]\n" op                      \
       " v[16:17], v[30:31], v[32:33], v[16:17]\n" op " v[18:19], v[32:33], v[22:23], v[18:19]\n" op                      \
       " v[34:35], v[38:39], v[40:41], v[34:35]\n" op " v[36:37], v[40:41], v[42:43], v[36:37]\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, unsigned long long *clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float seed = threadIdx.x * 1e-3f + 1.0f;
    asm volatile(
        "v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"
        "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
        "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n"
        "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
        "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
        "v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
        "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n"
        "v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n"
        "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"
        "s_mov_b64 s[40:41], exec\n s_mov_b64 s[42:43], 0\n s_mov_b64 s[44:45], exec\n s_mov_b64 s[46:47], 0\n"
        "s_mov_b64 s[48:49], exec\n s_mov_b64 s[50:51], 0\n s_mov_b64 s[52:53], exec\n s_mov_b64 s[54:55], 0\n"
        :
        : "v"(seed)
        : CLOB);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile("v_fma_f32 v8, v8, %0, 0.5\n v_fma_f32 v9, v9, %0, 0.5\n v_fma_f32 v10, v10, %0, 0.5\n v_fma_f32 v11, v11, %0, 0.5\n"
                         "v_fma_f32 v12, v12, %0, 0.5\n v_fma_f32 v13, v13, %0, 0.5\n v_fma_f32 v14, v14, %0, 0.5\n v_fma_f32 v15, v15, %0, 0.5\n"
                         ::"s"(a) : CLOB);
        if (MODE == 1) asm volatile(T2("v_add_f32")::: CLOB);
        if (MODE == 2) asm volatile(T2B("v_add_f32")::: CLOB);
        if (MODE == 3) asm volatile(T3("v_med3_f32")::: CLOB);
        if (MODE == 4) asm volatile(T3B("v_med3_f32")::: CLOB);
        if (MODE == 5) asm volatile(TCHAIN("v_med3_f32")::: CLOB);
        if (MODE == 6) asm volatile(T2("v_min_f32")::: CLOB);
        if (MODE == 7) asm volatile(T2("v_mul_f32")::: CLOB);
        if (MODE == 8) asm volatile(T3("v_fma_f32")::: CLOB);
        if (MODE == 9) asm volatile(T2("v_fmac_f32")::: CLOB);
        if (MODE == 10) asm volatile(T3("v_med3_i32")::: CLOB);
        if (MODE == 11) asm volatile(T3("v_med3_u32")::: CLOB);
        if (MODE == 12) asm volatile(TCHAIN("v_med3_u32")::: CLOB);
        if (MODE == 13) asm volatile(T2("v_min_u32")::: CLOB);
        if (MODE == 14) asm volatile(T2("v_max_u32")::: CLOB);
        if (MODE == 15) asm volatile(T2("v_min_i32")::: CLOB);
        if (MODE == 16) asm volatile(T3("v_min3_u32")::: CLOB);
        if (MODE == 17) asm volatile(T3("v_max3_f32")::: CLOB);
        if (MODE == 18) asm volatile(T2("v_and_b32")::: CLOB);
        if (MODE == 19) asm volatile(T2("v_or_b32")::: CLOB);
        if (MODE == 20) asm volatile(T3("v_and_or_b32")::: CLOB);
        if (MODE == 21) asm volatile(T3("v_bfi_b32")::: CLOB);
        if (MODE == 22) asm volatile(T3("v_perm_b32")::: CLOB);
        if (MODE == 23) asm volatile(T3("v_lshl_or_b32")::: CLOB);
        if (MODE == 24) asm volatile(T2("v_add_u32")::: CLOB);
        if (MODE == 25) asm volatile(T3("v_add3_u32")::: CLOB);
        if (MODE == 26) asm volatile(T2("v_sub_u32")::: CLOB);
        if (MODE == 27) asm volatile(TCMP("v_cmp_lt_f32")::: CLOB);
        if (MODE == 28) asm volatile(TCMP("v_cmp_lt_u32")::: CLOB);
        if (MODE == 29) asm volatile(TCND::: CLOB);
        if (MODE == 30) asm volatile(T2D("v_pk_mul_f32")::: CLOB);
        if (MODE == 31) asm volatile(T2D("v_pk_add_f32")::: CLOB);
        if (MODE == 32) asm volatile(T3D("v_pk_fma_f32")::: CLOB);
        if (MODE == 33) asm volatile(T2("v_pk_min_f16")::: CLOB);
        if (MODE == 34) asm volatile(T2("v_pk_max_u16")::: CLOB);
        if (MODE == 35) asm volatile(T2("v_pk_add_f16")::: CLOB);
        if (MODE == 36) asm volatile(T2D("v_add_f64")::: CLOB);
        if (MODE == 37) asm volatile(T2D("v_mul_f64")::: CLOB);
        if (MODE == 38) asm volatile(T3D("v_fma_f64")::: CLOB);
        if (MODE == 39) asm volatile(T2D("v_min_f64")::: CLOB);
        if (MODE == 40) asm volatile(T2("v_max_f32")::: CLOB);
        if (MODE == 41) asm volatile(T2("v_min_f16")::: CLOB);
        if (MODE == 42) asm volatile(T2("v_lshlrev_b32")::: CLOB);
        if (MODE == 43) asm volatile(T2("v_xor_b32")::: CLOB);
        if (MODE == 44) asm volatile(T3("v_mad_u32_u24")::: CLOB);
        if (MODE == 45) asm volatile(T3("v_alignbit_b32")::: CLOB);
        if (MODE == 46) asm volatile(T2("v_sub_f32")::: CLOB);
        if (MODE == 47) asm volatile(T3("v_sad_u32")::: CLOB);
        if (MODE == 48) asm volatile(T3("v_xad_u32")::: CLOB);
        if (MODE == 49) asm volatile(T3("v_max3_u32")::: CLOB);
    }
    float s;
    asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n"
                 "v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v15\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v17\n v_add_f32 %0, %0, v30\n"
                 : "=v"(s)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, float *d, unsigned long long *clk)
{
    const int iters = 20000;
    printf("%-28s", name);
    for (int blocks : {256, 512, 768, 1024, 2048}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memrealtime ticks at 100 MHz
        const double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
        const double ns = ms * 1e6 / per_simd;
        printf("  %dw: %5.2f cyc (%.2f ns, %.2f GHz)", blocks / 256, ns * ghz, ns, ghz);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    float *d;
    unsigned long long *clk;
    hipMalloc(&d, 256 * 8192 * 4);
    hipMalloc(&clk, 64);
    printf("cycles per wave-instruction per SIMD at 1, 2, 3, 4, 8 waves per SIMD (8 independent instructions per trip)\n");
    run<0>("v_fma_f32 v,v,s,const", d, clk);
    run<1>("v_add_f32 (2 banks)", d, clk);
    run<2>("v_add_f32 (1 bank)", d, clk);
    run<46>("v_sub_f32", d, clk);
    run<7>("v_mul_f32", d, clk);
    run<9>("v_fmac_f32", d, clk);
    run<8>("v_fma_f32 3 vgpr", d, clk);
    run<6>("v_min_f32", d, clk);
    run<40>("v_max_f32", d, clk);
    run<3>("v_med3_f32 (3 banks)", d, clk);
    run<4>("v_med3_f32 (1 bank)", d, clk);
    run<5>("v_med3_f32 insert chain", d, clk);
    run<17>("v_max3_f32", d, clk);
    run<10>("v_med3_i32", d, clk);
    run<11>("v_med3_u32", d, clk);
    run<12>("v_med3_u32 insert chain", d, clk);
    run<13>("v_min_u32", d, clk);
    run<14>("v_max_u32", d, clk);
    run<15>("v_min_i32", d, clk);
    run<16>("v_min3_u32", d, clk);
    run<49>("v_max3_u32", d, clk);
    run<18>("v_and_b32", d, clk);
    run<19>("v_or_b32", d, clk);
    run<43>("v_xor_b32", d, clk);
    run<42>("v_lshlrev_b32", d, clk);
    run<20>("v_and_or_b32", d, clk);
    run<21>("v_bfi_b32", d, clk);
    run<22>("v_perm_b32", d, clk);
    run<23>("v_lshl_or_b32", d, clk);
    run<45>("v_alignbit_b32", d, clk);
    run<24>("v_add_u32", d, clk);
    run<26>("v_sub_u32", d, clk);
    run<25>("v_add3_u32", d, clk);
    run<44>("v_mad_u32_u24", d, clk);
    run<47>("v_sad_u32", d, clk);
    run<48>("v_xad_u32", d, clk);
    run<27>("v_cmp_lt_f32 -> sgpr pair", d, clk);
    run<28>("v_cmp_lt_u32 -> sgpr pair", d, clk);
    run<29>("v_cndmask_b32 (sgpr pair)", d, clk);
    run<30>("v_pk_mul_f32", d, clk);
    run<31>("v_pk_add_f32", d, clk);
    run<32>("v_pk_fma_f32", d, clk);
    run<33>("v_pk_min_f16", d, clk);
    run<34>("v_pk_max_u16", d, clk);
    run<35>("v_pk_add_f16", d, clk);
    run<41>("v_min_f16", d, clk);
    run<36>("v_add_f64", d, clk);
    run<37>("v_mul_f64", d, clk);
    run<38>("v_fma_f64", d, clk);
    run<39>("v_min_f64", d, clk);
    return 0;
}
