// Does a wave's own vector work hide under its own MFMAs on gfx950, or only under ANOTHER wave's?
// (decides whether the recurrent cells / features extractor should run one wave per SIMD with the epilogue dealt out between the
// MFMAs -- what lstm_mfma64 / policy_fe64 / policy_rc64 do -- or two waves per SIMD in alternating MFMA / epilogue phases.)
// 256 workgroups; per iteration a wave issues 2 MFMAs (32x32x16 bf16, two independent chains) and / or NV vector instructions.
//   modes with 256 threads: one wave per SIMD.   modes with 512 threads: two waves per SIMD (waves w and w + 4 share a SIMD).
// build: hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o bin/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int ITERS = 1024;

#define MFMA_A() asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b))
#define MFMA_B() asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b))
#define FMA2()  asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(m), "v"(d))
#define EXP2()  asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(x4), "+v"(x5))
#define LDSR()  asm volatile("ds_read_b128 %0, %1" : "=v"(lw) : "v"(laddr))
#define MFMA2() asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b))
#define FMA4()  asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(d))
#define FMA4B() asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(d))
#define EXP4()  asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3))
#define EXP4B() asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))
#define PK4()   asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pd))

// MODE: 0 MFMA only | 1 MFMA + 8 fma | 2 MFMA + 8 exp | 3 8 fma only | 4 8 exp only | 5 MFMA + 4 fma | 6 MFMA + 16 fma
//       7 MFMA + 4 pk_fma | 8 MFMA + 4 exp | 9 MFMA + 4 exp + 4 fma (alternating, as an epilogue stage would)
//       10 (512 thr) waves 0-3 MFMA only, waves 4-7 16 fma       11 (512) waves 0-3 MFMA only, waves 4-7 8 exp + 8 fma
//       12 (512) every wave MFMA + 8 fma interleaved              13 (512) every wave: PH iterations MFMA-only then PH iterations
//       of 16 fma, the two waves of a SIMD in opposite phases, free running     14 as 13 with a workgroup barrier per phase
//       15 (512) as 13, epilogue = 8 exp + 8 fma per iteration
template <int MODE, int THREADS> __global__ void __launch_bounds__(THREADS, 1) k(float* out, uint64_t* cyc, const u32x4_t* src)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    u32x4_t a = src[threadIdx.x & 255], b = src[256 + (threadIdx.x & 255)];
    f32x16_t c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float m = 0.999f, d = 1e-3f;
    v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const v2 pm = {m, m}, pd = {d, d};
    const int wave = threadIdx.x >> 6;
    __shared__ u32x4_t lds[256];
    lds[threadIdx.x & 255] = a;
    u32x4_t lw = a; const uint32_t laddr = (threadIdx.x & 63) * 16;
    constexpr int PH = 12;                                  // iterations per phase: 24 MFMAs = one 32-row tile over K = 384
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE <= 9) {
#pragma unroll 4
        for (int i = 0; i < ITERS; ++i) {
            if constexpr (MODE == 0) { MFMA2(); }
            if constexpr (MODE == 1) { MFMA2(); FMA4(); FMA4B(); }
            if constexpr (MODE == 2) { MFMA2(); EXP4(); EXP4B(); }
            if constexpr (MODE == 3) { FMA4(); FMA4B(); }
            if constexpr (MODE == 4) { EXP4(); EXP4B(); }
            if constexpr (MODE == 5) { MFMA2(); FMA4(); }
            if constexpr (MODE == 6) { MFMA2(); FMA4(); FMA4B(); FMA4(); FMA4B(); }
            if constexpr (MODE == 7) { MFMA2(); PK4(); }
            if constexpr (MODE == 8) { MFMA2(); EXP4(); }
            if constexpr (MODE == 9) { MFMA2(); EXP4(); FMA4B(); }
        }
    } else if constexpr (MODE >= 20 && MODE < 40) {
#pragma unroll 4
        for (int i = 0; i < ITERS; ++i) {
            if constexpr (MODE == 20) { MFMA_A(); FMA2(); MFMA_B(); FMA2(); }
            if constexpr (MODE == 21) { MFMA_A(); FMA4(); MFMA_B(); FMA4B(); }
            if constexpr (MODE == 22) { MFMA_A(); FMA4(); FMA2(); MFMA_B(); FMA4B(); FMA2(); }
            if constexpr (MODE == 23) { MFMA_A(); FMA4(); FMA4B(); MFMA_B(); FMA4(); FMA4B(); }
            if constexpr (MODE == 24) { MFMA_A(); EXP2(); MFMA_B(); EXP2(); }
            if constexpr (MODE == 25) { MFMA_A(); EXP4(); MFMA_B(); EXP4B(); }
            if constexpr (MODE == 26) { MFMA_A(); EXP2(); FMA2(); MFMA_B(); EXP2(); FMA2(); }
            if constexpr (MODE == 27) { MFMA_A(); EXP4(); FMA4B(); MFMA_B(); EXP4(); FMA4B(); }
            if constexpr (MODE == 28) { MFMA_A(); LDSR(); FMA4(); MFMA_B(); FMA4B(); }
            if constexpr (MODE == 29) { MFMA_A(); LDSR(); MFMA_B(); }
            if constexpr (MODE == 30) { MFMA_A(); FMA4(); FMA4B(); FMA4(); MFMA_B(); FMA4B(); FMA4(); FMA4B(); }
            if constexpr (MODE == 31) { MFMA_A(); FMA4(); __builtin_amdgcn_sched_barrier(0); MFMA_B(); FMA4B(); __builtin_amdgcn_sched_barrier(0); }
        }
    } else if constexpr (MODE == 10 || MODE == 11) {
        if (wave < 4) {
#pragma unroll 4
            for (int i = 0; i < ITERS; ++i) MFMA2();
        } else {
#pragma unroll 4
            for (int i = 0; i < ITERS; ++i) {
                if constexpr (MODE == 10) { FMA4(); FMA4B(); FMA4(); FMA4B(); }
                else { EXP4(); FMA4B(); EXP4(); FMA4B(); }
            }
        }
    } else if constexpr (MODE == 12) {
#pragma unroll 4
        for (int i = 0; i < ITERS; ++i) { MFMA2(); FMA4(); FMA4B(); }
    } else {
        // alternating phases; wave w starts with MFMAs, wave w + 4 with the vector phase
        for (int ph = 0; ph < 2 * ITERS / PH; ++ph) {
            const bool mf = ((ph + (wave >> 2)) & 1) == 0;
            if (mf) {
#pragma unroll
                for (int i = 0; i < PH; ++i) MFMA2();
            } else {
#pragma unroll
                for (int i = 0; i < PH; ++i) {
                    if constexpr (MODE == 15) { EXP4(); FMA4B(); EXP4(); FMA4B(); }
                    else { FMA4(); FMA4B(); FMA4(); FMA4B(); }
                }
            }
            if constexpr (MODE == 14) __syncthreads();
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = __builtin_bit_cast(float, lw.x) + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    for (int e = 0; e < 16; ++e) acc += c0[e] + c1[e];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE, int THREADS> void run(float* out, uint64_t* cyc, const u32x4_t* src, const char* name, double mfma_per_simd, const char* note)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(cyc, 0, 2048 * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, src);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static uint64_t h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = THREADS / 64;
    double lo = 0, hi = 0;                                   // waves 0-3 and waves 4-7 apart
    for (int g = 0; g < 256; ++g) for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += double(h[g * 8 + w]);
    lo /= 1024.0; hi /= (nw > 4 ? 1024.0 : 1.0);
    printf("%-58s %7.1f", name, lo / ITERS);
    if (nw > 4) printf(" | %7.1f", hi / ITERS); else printf(" | %7s", "-");
    printf("   cycles per iteration (waves 0-3 | 4-7)   %7.1f us", ms * 100);
    if (mfma_per_simd > 0) printf("   %5.1f cycles of SIMD time per MFMA", (lo > hi ? lo : hi) / mfma_per_simd);
    printf("   %s\n", note);
}
int main()
{
    float* out; uint64_t* cyc; u32x4_t* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8); hipMalloc(&src, 1024 * 16);
    uint16_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = uint16_t(0x3c00 + (i * 2654435761u >> 22) % 0x300);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    printf("one iteration = 2 MFMAs and / or the vector instructions named; ITERS = %d\n", ITERS);
    run<0, 256>(out, cyc, src, " 0: 1 wave/SIMD  2 MFMA", 2.0 * ITERS, "");
    run<3, 256>(out, cyc, src, " 3: 1 wave/SIMD  8 fma", 0, "");
    run<4, 256>(out, cyc, src, " 4: 1 wave/SIMD  8 exp", 0, "");
    run<5, 256>(out, cyc, src, " 5: 1 wave/SIMD  2 MFMA + 4 fma", 2.0 * ITERS, "");
    run<1, 256>(out, cyc, src, " 1: 1 wave/SIMD  2 MFMA + 8 fma", 2.0 * ITERS, "");
    run<6, 256>(out, cyc, src, " 6: 1 wave/SIMD  2 MFMA + 16 fma", 2.0 * ITERS, "");
    run<7, 256>(out, cyc, src, " 7: 1 wave/SIMD  2 MFMA + 4 pk_fma", 2.0 * ITERS, "");
    run<8, 256>(out, cyc, src, " 8: 1 wave/SIMD  2 MFMA + 4 exp", 2.0 * ITERS, "");
    run<2, 256>(out, cyc, src, " 2: 1 wave/SIMD  2 MFMA + 8 exp", 2.0 * ITERS, "");
    run<9, 256>(out, cyc, src, " 9: 1 wave/SIMD  2 MFMA + 4 exp + 4 fma", 2.0 * ITERS, "");
    run<20, 256>(out, cyc, src, "20: MFMA, 2 fma, MFMA, 2 fma", 2.0 * ITERS, "");
    run<21, 256>(out, cyc, src, "21: MFMA, 4 fma, MFMA, 4 fma", 2.0 * ITERS, "");
    run<22, 256>(out, cyc, src, "22: MFMA, 6 fma, MFMA, 6 fma", 2.0 * ITERS, "");
    run<23, 256>(out, cyc, src, "23: MFMA, 8 fma, MFMA, 8 fma", 2.0 * ITERS, "");
    run<30, 256>(out, cyc, src, "30: MFMA, 12 fma, MFMA, 12 fma", 2.0 * ITERS, "");
    run<24, 256>(out, cyc, src, "24: MFMA, 2 exp, MFMA, 2 exp", 2.0 * ITERS, "");
    run<25, 256>(out, cyc, src, "25: MFMA, 4 exp, MFMA, 4 exp", 2.0 * ITERS, "");
    run<26, 256>(out, cyc, src, "26: MFMA, 2 exp + 2 fma, MFMA, 2 exp + 2 fma", 2.0 * ITERS, "");
    run<27, 256>(out, cyc, src, "27: MFMA, 4 exp + 4 fma, MFMA, 4 exp + 4 fma", 2.0 * ITERS, "");
    run<28, 256>(out, cyc, src, "28: MFMA, ds_read_b128 + 4 fma, MFMA, 4 fma", 2.0 * ITERS, "");
    run<29, 256>(out, cyc, src, "29: MFMA, ds_read_b128, MFMA", 2.0 * ITERS, "");
    run<31, 256>(out, cyc, src, "31: as 21 with sched_barrier(0) after each group", 2.0 * ITERS, "");
    run<10, 512>(out, cyc, src, "10: 2 waves/SIMD one 2 MFMA, the other 16 fma", 2.0 * ITERS, "");
    run<11, 512>(out, cyc, src, "11: 2 waves/SIMD one 2 MFMA, the other 8 exp + 8 fma", 2.0 * ITERS, "");
    run<12, 512>(out, cyc, src, "12: 2 waves/SIMD both 2 MFMA + 8 fma", 4.0 * ITERS, "");
    run<13, 512>(out, cyc, src, "13: 2 waves/SIMD alternating 24-MFMA / 192-fma phases", 4.0 * ITERS, "(each wave: ITERS/12 phases of each kind; per-iteration = per 2 MFMA of the PAIR)");
    run<14, 512>(out, cyc, src, "14: as 13 + barrier per phase", 4.0 * ITERS, "");
    run<15, 512>(out, cyc, src, "15: as 13, vector phase = 96 exp + 96 fma", 4.0 * ITERS, "");
    return 0;
}
