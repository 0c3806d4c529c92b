// Single-wave VALU issue costs on gfx950: what does ONE wave per SIMD sustain for scalar-f32, packed-f32 and
// dependent chains?  (decides whether packing two independent quantities per instruction pays in the env-step kernel)
// build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue ; run: ./issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
#define ITERS 200

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, uint64_t* cyc)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 1e-7f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2 pb = {b, b}, pc = {c, c};
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {          // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 1) {   // dependent chain of 8 v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (MODE == 2) {   // 8 independent v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
            } else if (MODE == 3) {   // dependent chain of 8 v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(p0) : "v"(pb), "v"(pc));
            } else if (MODE == 4) {   // 2 interleaved dependent chains of v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                             : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
            } else if (MODE == 5) {   // 8 independent v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
            } else if (MODE == 6) {   // 8 independent v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 7) {   // 8 independent v_fma_f64
                double d0 = a0, d1 = a1;
                asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                             "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                             : "+v"(d0), "+v"(d1) : "v"((double)b), "v"((double)c));
                a0 = (float)d0; a1 = (float)d1;
            } else if (MODE == 8) {   // 4 independent chains (scalar), 2 deep
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            } else if (MODE == 10) {  // 8 independent v_mul_f32 in the 4-byte VOP2 encoding (the env step's hot loop is half e32 forms)
                asm volatile("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                             "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 11) {  // 8 independent v_fmac_f32 (VOP2, 4 bytes)
                asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                             "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 12) {  // v_fmaak_f32 (VOP2 + 32-bit literal, 8 bytes)
                asm volatile("v_fmaak_f32 %0, %8, %0, 0x3a83126f\n v_fmaak_f32 %1, %8, %1, 0x3a83126f\n v_fmaak_f32 %2, %8, %2, 0x3a83126f\n v_fmaak_f32 %3, %8, %3, 0x3a83126f\n"
                             "v_fmaak_f32 %4, %8, %4, 0x3a83126f\n v_fmaak_f32 %5, %8, %5, 0x3a83126f\n v_fmaak_f32 %6, %8, %6, 0x3a83126f\n v_fmaak_f32 %7, %8, %7, 0x3a83126f\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (MODE == 9) {   // v_med3_f32 + v_cndmask mix: 8 independent v_med3
                asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                             "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE> void run(const char* name, int threads, float* out, uint64_t* cyc, uint64_t* h)
{
    const int blocks = 256;
    k<MODE><<<blocks, threads>>>(out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int waves = blocks * threads / 64;
    hipMemcpy(h, cyc, waves * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < waves; ++i) sum += h[i];
    double per = sum / waves / (double)(ITERS * REP);
    printf("%-44s waves/SIMD %d : %6.2f cycles(s_memtime)/instr/wave   kernel %.1f us\n", name, threads / 256, per, ms * 1e3);
}

int main()
{
    float* out; uint64_t* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    uint64_t* h = (uint64_t*)malloc(256 * 8 * 8);
    for (int t = 256; t <= 512; t += 256) {
        run<0>("8 independent v_fma_f32", t, out, cyc, h);
        run<1>("dependent v_fma_f32 chain", t, out, cyc, h);
        run<4>("2 interleaved dependent v_fma_f32 chains", t, out, cyc, h);
        run<8>("4 interleaved dependent v_fma_f32 chains", t, out, cyc, h);
        run<2>("8 independent v_pk_fma_f32", t, out, cyc, h);
        run<3>("dependent v_pk_fma_f32 chain", t, out, cyc, h);
        run<5>("8 independent v_pk_mul_f32", t, out, cyc, h);
        run<6>("8 independent v_rcp_f32", t, out, cyc, h);
        run<7>("2 interleaved dependent v_fma_f64 chains", t, out, cyc, h);
        run<9>("8 independent v_med3_f32", t, out, cyc, h);
        run<10>("8 independent v_mul_f32_e32 (4-byte)", t, out, cyc, h);
        run<11>("8 independent v_fmac_f32_e32 (4-byte)", t, out, cyc, h);
        run<12>("8 independent v_fmaak_f32 (4 + 4 bytes)", t, out, cyc, h);
    }
    return 0;
}
