// policy_trunk.hip -- the two trunks of the policy's mlp_extractor as ONE launch:
//
//   h_pi [B][256] bf16 -> Linear(256,128)+ReLU -> Linear(128,64)+ReLU -> lat_pi [B][64] bf16      (blockIdx.y = 0)
//   h_vf [B][256] bf16 -> Linear(256,128)+ReLU -> Linear(128,64)+ReLU -> lat_vf [B][64] bf16      (blockIdx.y = 1)
//
// (sb3_contrib's MlpLstmPolicy with the reference's net_arch pi = vf = [128, 64],
// learned_controllers/networks/lstm_policy.py:107-136; the 64 -> 4 / 64 -> 1 output layers and the sampling are
// fdyn_policy_heads.)  As four hipBLASLt GEMMs these layers took 2 x (16 + 10) us of a 300 us rollout step at 65 536 envs and
// moved the 128-wide intermediate through HBM twice; the arithmetic is 10.7 GFLOP, the traffic that has to exist 84 MB.
//
// Scheme (the operand roles of policy_fe64.hip, compiler-managed registers): the WEIGHTS are the MFMA A operand, read from LDS
// (both layers of a trunk stay resident: 66 + 17 KB), the ACTIVATIONS are the B operand in registers.  A 32x32 output tile is
// then D[n][b]: lane (b, hf) holds, for ITS OWN batch row b, 16 features n = (e & 3) + 8 (e >> 2) + 4 hf of the tile; elements
// 8 q .. 8 q + 7, with bias and ReLU applied and rounded to bf16, ARE the B fragment of the next layer's k-step 2 tile + q --
// no transposition, no LDS, no HBM between the layers.  The price is a fixed order of k inside every block of 16 of layer 2,
// (0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15), which the host folds into the weight image (policy.py: _KPERM16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fdyn.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int K1 = 256, N1 = 128, N2 = 64;
constexpr int PADE = 8;                                  // bf16 elements (16 B) of LDS row padding: conflict-free ds_read_b128
constexpr int ROW1 = K1 + PADE, ROW2 = N1 + PADE;
constexpr int TRUNK_WGS = 128;                           // workgroups per trunk: 2 x 128 = one per CU on MI355X

__global__ void __launch_bounds__(256, 1)
policy_trunk_kernel(const uint16_t* __restrict__ h_pi, const uint16_t* __restrict__ h_vf /*[B][256] bf16*/,
                    const uint16_t* __restrict__ W1 /*[2][128][256] bf16*/, const float* __restrict__ b1 /*[2][128]*/,
                    const uint16_t* __restrict__ W2p /*[2][64][128] bf16, k permuted per block of 16*/,
                    const float* __restrict__ b2 /*[2][64]*/, uint16_t* __restrict__ lat_pi, uint16_t* __restrict__ lat_vf /*[B][64] bf16*/,
                    int64_t B, int64_t rows_per_wg)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_w1[N1 * ROW1];
    __shared__ __attribute__((aligned(16))) uint16_t s_w2[N2 * ROW2];
    __shared__ __attribute__((aligned(16))) float s_b1[N1];
    __shared__ __attribute__((aligned(16))) float s_b2[N2];
    // a wave's 32 input rows, staged with fully coalesced 16-byte loads (a wave instruction covers two whole rows).  Read straight
    // into the B fragments, lane (b, hf) takes 16 bytes of ITS row per instruction: 32 rows x 32 B per instruction, every 128-byte
    // line fetched four times -- and the 64 KB a workgroup has in flight do not fit the 32 KB L1, so the re-fetches go to L2
    // (first version of this kernel: 40 us against 51 for the four GEMMs it replaces)
    __shared__ __attribute__((aligned(16))) uint16_t s_h[4][32 * ROW1];
    const int trunk = blockIdx.y;
    const uint16_t* __restrict__ h = trunk ? h_vf : h_pi;
    uint16_t* __restrict__ lat = trunk ? lat_vf : lat_pi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;

    // ---- this trunk's weights -> LDS (16-byte pieces; every workgroup reads the same 100 KB: L2 hits)
    // all 20 loads of a thread in flight together, then the LDS writes: as a load -> store loop the staging was 20 dependent
    // L2 round trips, ~30 of the first version's 40 us
    {
        constexpr int P1 = N1 * (K1 / 8) / 256, P2 = N2 * (N1 / 8) / 256;          // 16 and 4 pieces per thread
        static_assert(N1 * (K1 / 8) % 256 == 0 && N2 * (N1 / 8) % 256 == 0, "whole pieces per thread");
        uint4 w1r[P1], w2r[P2];
#pragma unroll
        for (int i = 0; i < P1; ++i) {
            const int v = tid + 256 * i, row = v / (K1 / 8), c = v % (K1 / 8);
            w1r[i] = *reinterpret_cast<const uint4*>(W1 + (int64_t(trunk) * N1 + row) * K1 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < P2; ++i) {
            const int v = tid + 256 * i, row = v / (N1 / 8), c = v % (N1 / 8);
            w2r[i] = *reinterpret_cast<const uint4*>(W2p + (int64_t(trunk) * N2 + row) * N1 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < P1; ++i) {
            const int v = tid + 256 * i, row = v / (K1 / 8), c = v % (K1 / 8);
            *reinterpret_cast<uint4*>(s_w1 + row * ROW1 + c * 8) = w1r[i];
        }
#pragma unroll
        for (int i = 0; i < P2; ++i) {
            const int v = tid + 256 * i, row = v / (N1 / 8), c = v % (N1 / 8);
            *reinterpret_cast<uint4*>(s_w2 + row * ROW2 + c * 8) = w2r[i];
        }
    }
    if (tid < N1) s_b1[tid] = b1[trunk * N1 + tid];
    if (tid < N2) s_b2[tid] = b2[trunk * N2 + tid];
    __syncthreads();

    const int64_t wg_first = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t wg_end = wg_first + rows_per_wg < B ? wg_first + rows_per_wg : B;
    for (int64_t row0 = wg_first + wave * 32; row0 < wg_end; row0 += 128) {          // wave-uniform
        const int64_t my_row = row0 + r;
        const int64_t lrow = my_row < B ? my_row : B - 1;                              // rows past the end read the last row, store nothing
        // ---- the wave's 32 rows -> LDS (piece p = 64 i + lane: row p / 32, 16-byte column p % 32)
        (void)lrow;
        uint16_t* sh = s_h[wave];
        {
            uint4 pieces[K1 / 16];
#pragma unroll
            for (int i = 0; i < K1 / 16; ++i) {
                const int pc = 64 * i + lane, prow = pc >> 5, pcol = pc & 31;
                const int64_t src = row0 + prow < B ? row0 + prow : B - 1;             // rows past the end read the last row
                pieces[i] = *reinterpret_cast<const uint4*>(h + src * K1 + pcol * 8);
            }
            // (a wave's LDS operations execute in program order: the previous iteration's fragment reads precede these writes,
            // the reads below follow them -- no barrier, the region is this wave's own)
#pragma unroll
            for (int i = 0; i < K1 / 16; ++i) {
                const int pc = 64 * i + lane, prow = pc >> 5, pcol = pc & 31;
                *reinterpret_cast<uint4*>(sh + prow * ROW1 + pcol * 8) = pieces[i];
            }
        }
        // ---- layer-1 B fragments: lane (b, hf) holds h[b][16 s + 8 hf .. + 7] for every k-step s
        bf16x8_t bx[K1 / 16];
        const uint16_t* hr = sh + r * ROW1 + 8 * hf;
#pragma unroll
        for (int s = 0; s < K1 / 16; ++s) bx[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(hr + 16 * s));

        // ---- layer 1: four 32-feature tiles; each leaves two B fragments of layer 2
        bf16x8_t b2f[N1 / 16];
#pragma unroll
        for (int t = 0; t < N1 / 32; ++t) {
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const uint16_t* wr = s_w1 + (32 * t + r) * ROW1 + 8 * hf;
#pragma unroll
            for (int s = 0; s < K1 / 16; ++s) {
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wr + 16 * s));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8_t f;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * q + jj;                                          // e >> 2
                    const float4 bb = *reinterpret_cast<const float4*>(s_b1 + 32 * t + 8 * j + 4 * hf);
                    const float bv[4] = { bb.x, bb.y, bb.z, bb.w };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc[4 * j + i] + bv[i];
                        f[4 * jj + i] = static_cast<__bf16>(v > 0.0f ? v : 0.0f);
                    }
                }
                b2f[2 * t + q] = f;
            }
        }

        // ---- layer 2: two 32-feature tiles, stored as 8-byte pieces (4 adjacent features of this lane's own row)
#pragma unroll
        for (int t = 0; t < N2 / 32; ++t) {
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const uint16_t* wr = s_w2 + (32 * t + r) * ROW2 + 8 * hf;
#pragma unroll
            for (int s = 0; s < N1 / 16; ++s) {
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wr + 16 * s));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2f[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bb = *reinterpret_cast<const float4*>(s_b2 + 32 * t + 8 * j + 4 * hf);
                const float bv[4] = { bb.x, bb.y, bb.z, bb.w };
                bf16x4_t o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[4 * j + i] + bv[i];
                    o[i] = static_cast<__bf16>(v > 0.0f ? v : 0.0f);
                }
                if (my_row < B) *reinterpret_cast<uint2*>(lat + my_row * N2 + 32 * t + 8 * j + 4 * hf) = __builtin_bit_cast(uint2, o);
            }
        }
    }
}

}  // namespace

extern "C" int fdyn_policy_trunks(const void* h_pi, const void* h_vf, const void* W1, const float* b1, const void* W2p, const float* b2,
                                  void* lat_pi, void* lat_vf, int64_t B, void* stream)
{
    if (B < 0) return FDYN_ERR_BAD_SIZE;
    if (!h_pi || !h_vf || !W1 || !b1 || !W2p || !b2 || !lat_pi || !lat_vf) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const int64_t blocks128 = (B + 127) / 128;
    const int64_t gx = blocks128 < TRUNK_WGS ? blocks128 : TRUNK_WGS;
    const int64_t rows_per_wg = ((blocks128 + gx - 1) / gx) * 128;
    hipLaunchKernelGGL(policy_trunk_kernel, dim3(unsigned(gx), 2), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)h_pi,
                       (const uint16_t*)h_vf, (const uint16_t*)W1, b1, (const uint16_t*)W2p, b2, (uint16_t*)lat_pi, (uint16_t*)lat_vf, B,
                       rows_per_wg);
    return int(hipGetLastError());
}
