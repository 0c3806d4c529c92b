// Which operand placement sustains the matrix pipe at ONE wave per SIMD?  v_mfma_f32_32x32x16_bf16 back to back, two
// independent accumulation chains (as policy_rc64 / policy_fe64 run them), 256 workgroups x 4 waves, N MFMAs per wave:
//   0: A, B, C/D all in architectural VGPRs          1: B operand in accumulator registers (rc64 / fe64)
//   2: A operand in accumulator registers (mfma64)   3: C/D in accumulator registers, A and B in VGPRs
//   4: C/D and B in accumulator registers            5: as 1 with a ds_read_b128 per MFMA pair (weights from LDS)
// prints shader cycles per MFMA (s_memtime) and the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int PAIRS = 2048;   // MFMA pairs per wave
template <int MODE> __global__ void __launch_bounds__(256, 1) k(float* out, uint64_t* cyc, const u32x4_t* src)
{
    __shared__ u32x4_t s[256 * 4];
    for (int i = threadIdx.x; i < 1024; i += 256) s[i] = src[i];
    __syncthreads();
    asm volatile("" ::: "a0", "a95");
    u32x4_t a = src[threadIdx.x], b = src[256 + threadIdx.x];
    f32x16_t c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
    asm volatile("v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %1\n v_accvgpr_write_b32 a6, %2\n v_accvgpr_write_b32 a7, %3" :: "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
    for (int r = 32; r < 96; ++r) asm volatile("s_nop 0");
    if (MODE == 3 || MODE == 4) for (int i = 0; i < 1; ++i) {
        asm volatile("v_accvgpr_write_b32 a32, 0\n v_accvgpr_write_b32 a33, 0\n v_accvgpr_write_b32 a34, 0\n v_accvgpr_write_b32 a35, 0\n v_accvgpr_write_b32 a36, 0\n v_accvgpr_write_b32 a37, 0\n v_accvgpr_write_b32 a38, 0\n v_accvgpr_write_b32 a39, 0\n"
                     "v_accvgpr_write_b32 a40, 0\n v_accvgpr_write_b32 a41, 0\n v_accvgpr_write_b32 a42, 0\n v_accvgpr_write_b32 a43, 0\n v_accvgpr_write_b32 a44, 0\n v_accvgpr_write_b32 a45, 0\n v_accvgpr_write_b32 a46, 0\n v_accvgpr_write_b32 a47, 0\n"
                     "v_accvgpr_write_b32 a48, 0\n v_accvgpr_write_b32 a49, 0\n v_accvgpr_write_b32 a50, 0\n v_accvgpr_write_b32 a51, 0\n v_accvgpr_write_b32 a52, 0\n v_accvgpr_write_b32 a53, 0\n v_accvgpr_write_b32 a54, 0\n v_accvgpr_write_b32 a55, 0\n"
                     "v_accvgpr_write_b32 a56, 0\n v_accvgpr_write_b32 a57, 0\n v_accvgpr_write_b32 a58, 0\n v_accvgpr_write_b32 a59, 0\n v_accvgpr_write_b32 a60, 0\n v_accvgpr_write_b32 a61, 0\n v_accvgpr_write_b32 a62, 0\n v_accvgpr_write_b32 a63, 0");
    }
    const u32x4_t* sl = s + (threadIdx.x & 63);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
    for (int i = 0; i < PAIRS; ++i) {
        if (MODE == 0) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, a[4:7], %0\n v_mfma_f32_32x32x16_bf16 %1, %2, a[0:3], %1" : "+v"(c0), "+v"(c1) : "v"(a));
        } else if (MODE == 2) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[4:7], %2, %0\n v_mfma_f32_32x32x16_bf16 %1, a[0:3], %2, %1" : "+v"(c0), "+v"(c1) : "v"(a));
        } else if (MODE == 3) {
            asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]\n v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" :: "v"(a), "v"(b));
        } else if (MODE == 4) {
            asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, a[4:7], a[32:47]\n v_mfma_f32_32x32x16_bf16 a[48:63], %0, a[0:3], a[48:63]" :: "v"(a));
        } else {
            const u32x4_t w = sl[(i & 15) * 64];
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, a[4:7], %0\n v_mfma_f32_32x32x16_bf16 %1, %2, a[0:3], %1" : "+v"(c0), "+v"(c1) : "v"(a));
            a = w;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += c0[e] + c1[e];
    if (MODE == 3 || MODE == 4) { float v; asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(v)); acc += v; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(float* out, uint64_t* cyc, const u32x4_t* src, const char* name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, src);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 1024; ++i) m += double(h[i]); m /= 1024;
    printf("%-44s %6.1f shader cycles per MFMA   %7.1f us per launch   %6.0f TFLOP/s\n", name, m / (2.0 * PAIRS), ms * 100, 1024.0 * 2 * PAIRS * 2 * 32 * 32 * 16 / (ms * 1e-4) / 1e12);
}
int main()
{
    float* out; uint64_t* cyc; u32x4_t* src;
    hipMalloc(&out, 65536 * 4); hipMalloc(&cyc, 1024 * 8); hipMalloc(&src, 1024 * 16);
    uint16_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = uint16_t(0x3c00 + (i * 2654435761u >> 22) % 0x300);   // random-ish bf16 in [~0.008, ~0.06]
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(out, cyc, src, "0: A, B, C/D in VGPRs");
    run<1>(out, cyc, src, "1: B in AGPRs (rc64, fe64)");
    run<2>(out, cyc, src, "2: A in AGPRs (mfma64)");
    run<3>(out, cyc, src, "3: C/D in AGPRs");
    run<4>(out, cyc, src, "4: C/D and B in AGPRs");
    run<5>(out, cyc, src, "5: as 1 + one ds_read_b128 per MFMA pair");
    return 0;
}
