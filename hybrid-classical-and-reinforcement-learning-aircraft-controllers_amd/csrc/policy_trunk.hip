// policy_trunk.hip -- the two trunks of the policy's mlp_extractor as ONE launch:
//
//   h_pi [B][256] bf16 -> Linear(256,128)+ReLU -> Linear(128,64)+ReLU -> lat_pi [B][64] bf16      (blockIdx.y = 0)
//   h_vf [B][256] bf16 -> Linear(256,128)+ReLU -> Linear(128,64)+ReLU -> lat_vf [B][64] bf16      (blockIdx.y = 1)
//
// (sb3_contrib's MlpLstmPolicy with the reference's net_arch pi = vf = [128, 64],
// learned_controllers/networks/lstm_policy.py:107-136; the 64 -> 4 / 64 -> 1 output layers and the sampling are
// fdyn_policy_heads.)  As four hipBLASLt GEMMs these layers took 2 x (16 + 10) us of a 300 us rollout step at 65 536 envs and
// moved the 128-wide intermediate through HBM twice; the arithmetic is 10.7 GFLOP, the traffic that has to exist 84 MB.
// Measured (rocprofv3, inside the rollout graph): 30.5 us -- first version 40.5 (weights staged by a load -> store loop: 20
// dependent L2 round trips), 34 with the next tile of rows prefetched, 30.5 with the A fragments requested a k-step ahead.
// Round 3, with the output heads and the sampling inside (HEADS): 33.3 -> 28.0 (heads as a third MFMA layer) -> 25.0 (EIGHT waves
// per workgroup = two per SIMD: a wave's tile is one dependent chain rows -> layer 1 -> layer 2 -> heads, and with a single wave
// on the SIMD nothing runs beside it; per-wave stamps scratch/bench_trunk_stamps.py, profiles/r03_trunk_phase_stamps.log).
//
// Scheme (the operand roles of policy_fe64.hip, compiler-managed registers): the WEIGHTS are the MFMA A operand, read from LDS
// (both layers of a trunk stay resident: 66 + 17 KB), the ACTIVATIONS are the B operand in registers.  A 32x32 output tile is
// then D[n][b]: lane (b, hf) holds, for ITS OWN batch row b, 16 features n = (e & 3) + 8 (e >> 2) + 4 hf of the tile; elements
// 8 q .. 8 q + 7, with bias and ReLU applied and rounded to bf16, ARE the B fragment of the next layer's k-step 2 tile + q --
// no transposition, no LDS, no HBM between the layers.  The price is a fixed order of k inside every block of 16 of layer 2,
// (0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15), which the host folds into the weight image (policy.py: _KPERM16).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/fdyn.h"
#include "philox.hpp"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));      // 16-byte piece (HIP's uint4 is a struct of unions: arrays of it stay in memory)

constexpr int K1 = 256, N1 = 128, N2 = 64;
constexpr int PADE = 8;                                  // bf16 elements (16 B) of LDS row padding: conflict-free ds_read_b128
constexpr int ROW1 = K1 + PADE, ROW2 = N1 + PADE;
constexpr int TRUNK_WGS = 128;                           // workgroups per trunk: 2 x 128 = one per CU on MI355X
constexpr int NT = 512;                                  // threads per workgroup: EIGHT waves = two per SIMD (see the kernel's head)
constexpr int KH = K1 / 2, ROWS = KH + PADE;             // a wave stages its 32 input rows half a row (128 columns) at a time

// compile-time loop: every index is a constant whatever the unroller decides (an array indexed by a loop the compiler leaves
// rolled lives in scratch memory)
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// output heads fused behind the trunks (HEADS): what fdyn_policy_heads does in a second launch -- mean = lat_pi Wa^T + ba,
// value = lat_vf wv + bv, Gaussian sampling + log-probability -- on the bf16-rounded trunk outputs while they are still in
// registers: a lane holds 32 of its row's 64 features, its partner lane (b, 1 - hf) the other 32.
struct HeadArgs {
    const uint16_t* Wa; const uint16_t* ba; const uint16_t* wv; const uint16_t* bv;      // [4][64], [4], [64], [1] bf16
    const float* log_std; uint64_t seed; const uint32_t* step; int deterministic;
    float* actions; float* logp; float* value;                                         // [B][4], [B], [B]
};

template <bool HEADS>
__global__ void __launch_bounds__(NT, 1)
policy_trunk_kernel(const uint16_t* __restrict__ h_pi, const uint16_t* __restrict__ h_vf /*[B][256] bf16*/,
                    const uint16_t* __restrict__ W1 /*[2][128][256] bf16*/, const float* __restrict__ b1 /*[2][128]*/,
                    const uint16_t* __restrict__ W2p /*[2][64][128] bf16, k permuted per block of 16*/,
                    const float* __restrict__ b2 /*[2][64]*/, uint16_t* __restrict__ lat_pi, uint16_t* __restrict__ lat_vf /*[B][64] bf16*/,
                    int64_t B, int64_t rows_per_wg, HeadArgs hd)
{
    // this trunk's head as a THIRD layer on the matrix pipe (round 3; as per-lane dot products -- 128 LDS words + 128 FMAs + 4
    // shuffles per lane and tile -- the heads were a third of the kernel, profiles/r03_trunk_phase_stamps.log): 32 output rows of
    // which 4 (pi: action_net) or 1 (vf: value_net) are real, K = 64 in the k order of layer 2's output fragments
    constexpr int ROWH = N2 + PADE;
    __shared__ __attribute__((aligned(16))) uint16_t s_hw[HEADS ? 32 * ROWH : 8];
    __shared__ float s_hb[4];                           // head biases
    __shared__ __attribute__((aligned(16))) uint16_t s_w1[N1 * ROW1];
    __shared__ __attribute__((aligned(16))) uint16_t s_w2[N2 * ROW2];
    __shared__ __attribute__((aligned(16))) float s_b1[N1];
    __shared__ __attribute__((aligned(16))) float s_b2[N2];
    // a wave's 32 input rows, staged with fully coalesced 16-byte loads (a wave instruction covers two whole rows).  Read straight
    // into the B fragments, lane (b, hf) takes 16 bytes of ITS row per instruction: 32 rows x 32 B per instruction, every 128-byte
    // line fetched four times -- and the 64 KB a workgroup has in flight do not fit the 32 KB L1, so the re-fetches go to L2
    // (first version of this kernel: 40 us against 51 for the four GEMMs it replaces)
    __shared__ __attribute__((aligned(16))) uint16_t s_h[NT / 64][32 * ROWS];
    const int trunk = blockIdx.y;
    const uint16_t* __restrict__ h = trunk ? h_vf : h_pi;
    uint16_t* __restrict__ lat = trunk ? lat_vf : lat_pi;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;

    const int64_t wg_first = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t wg_end = wg_first + rows_per_wg < B ? wg_first + rows_per_wg : B;
    // A wave's tile of 32 input rows travels HBM -> registers -> LDS -> B fragments, half a row at a time (the LDS that two waves
    // per SIMD leave beside the weights holds 32 x 128 columns per wave), and the NEXT tile's loads are issued as soon as the
    // current half has been written to LDS (piece p = 64 i + lane of a half: row p / 16, 16-byte column p % 16 -- a wave
    // instruction covers four half rows = eight whole 128-byte lines; rows past the end read the last row).
    // The first tiles are requested behind the staging loads, in front of their LDS writes: they stay in flight across the barrier.
    // (Staggered -- waves 4-7 requesting theirs after the barrier so that waves 0-3 compute meanwhile -- 25.0 us against 25.3.)
    u32x4_t pieces[2][KH / 16];
    auto request = [&](int64_t first, auto HALF) {
        constexpr int half = decltype(HALF)::value;
        sfor<0, KH / 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int pc = 64 * i + lane, prow = pc >> 4, pcol = pc & 15;
            const int64_t src = first + prow < B ? first + prow : B - 1;
            pieces[half][i] = *reinterpret_cast<const u32x4_t*>(h + src * K1 + half * KH + pcol * 8);
        });
    };
    using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
    // ---- this trunk's weights -> LDS (16-byte pieces; every workgroup reads the same 100 KB: L2 hits)
    // all 20 loads of a thread in flight together, then the LDS writes: as a load -> store loop the staging was 20 dependent
    // L2 round trips, ~30 of the first version's 40 us.
    constexpr int P1 = N1 * (K1 / 8) / NT, P2 = N2 * (N1 / 8) / NT;            // 8 and 2 pieces per thread
    static_assert(N1 * (K1 / 8) % NT == 0 && N2 * (N1 / 8) % NT == 0, "whole pieces per thread");
    u32x4_t w1r[P1], w2r[P2];
    // (every workgroup starting somewhere else in the image, so that 128 of them do not ask for the same lines in the same order:
    // no difference, 26.2 against 25.8 us)
#pragma unroll
    for (int i = 0; i < P1; ++i) {
        const int v = tid + NT * i, row = v / (K1 / 8), c = v % (K1 / 8);
        w1r[i] = *reinterpret_cast<const u32x4_t*>(W1 + (int64_t(trunk) * N1 + row) * K1 + c * 8);
    }
#pragma unroll
    for (int i = 0; i < P2; ++i) {
        const int v = tid + NT * i, row = v / (N1 / 8), c = v % (N1 / 8);
        w2r[i] = *reinterpret_cast<const u32x4_t*>(W2p + (int64_t(trunk) * N2 + row) * N1 + c * 8);
    }
    // the head image and the biases the same way: every load of the staging is in flight before the first wait (as load -> store
    // loops they were six dependent L2 round trips in front of the barrier), unconditional loads from clamped addresses, selected
    // afterwards (behind a branch the compiler cannot count what is in flight and waits for everything, the rows included)
    static_assert(4 * ROWH <= NT, "one head-weight load per thread");
    uint16_t hwr = 0, hbr = 0;
    const int hrow = tid / ROWH, hc = tid % ROWH;
    const bool hreal = hc < N2 && (trunk == 0 ? hrow < 4 : hrow == 0);
    if constexpr (HEADS) {
        // column c of the image = input feature 16 (c / 16) + _KPERM16[c % 16] (policy.py): the k order of layer 2's output fragments
        const int hsrc = 16 * ((hc & 63) / 16) + (hc & 3) + 4 * ((hc >> 3) & 1) + 8 * ((hc >> 2) & 1);
        hwr = (trunk == 0 ? hd.Wa : hd.wv)[hreal ? hrow * N2 + hsrc : 0];
        hbr = (trunk == 0 ? hd.ba : hd.bv)[trunk == 0 && tid < 4 ? tid : 0];
    }
    const float b1v = b1[trunk * N1 + (tid < N1 ? tid : 0)], b2v = b2[trunk * N2 + (tid < N2 ? tid : 0)];
    // The first rows are requested behind the staging loads and in front of their LDS writes: they stay in flight across the
    // barrier.  The CU's vector-memory path takes a 1 KB wave instruction every ~40 cycles here (per-wave stamps: the 26 loads of a
    // wave have been issued after 3.4 k cycles in the first four waves and after 8.5 k in the other four), so the order of the requests
    // is the order of arrival.  (Waves 4-7 requesting their rows later, so that the two waves of a SIMD start apart: 25.0 against
    // 25.3 us behind the barrier; behind a short s_sleep or behind the small staging, the compiler waits for staged data in
    // front of the row requests or sinks the requests below the barrier -- not pursued.  A bare s_barrier between the staging loads
    // and the row requests, so that no wave's weights queue behind another wave's rows: 25.7 against 25.8 us.)
    asm volatile("" ::: "memory");                     // the staging loads first: their wait must not cover a row load
    request(wg_first + wave * 32, H0{}); request(wg_first + wave * 32, H1{});       // rows past the end are clamped: no branch
    asm volatile("" ::: "memory");
    if constexpr (HEADS) {
#pragma unroll
        for (int j = 0; j < (32 * ROWH + NT - 1) / NT; ++j)
            if (tid + NT * j < 32 * ROWH) s_hw[tid + NT * j] = j == 0 && hreal ? hwr : uint16_t(0);
        if (tid < 4) s_hb[tid] = (trunk == 0 || tid == 0) ? __uint_as_float(uint32_t(hbr) << 16) : 0.0f;
    }
    if (tid < N1) s_b1[tid] = b1v;
    if (tid < N2) s_b2[tid] = b2v;
#pragma unroll
    for (int i = 0; i < P1; ++i) {
        const int v = tid + NT * i, row = v / (K1 / 8), c = v % (K1 / 8);
        *reinterpret_cast<u32x4_t*>(s_w1 + row * ROW1 + c * 8) = w1r[i];
    }
#pragma unroll
    for (int i = 0; i < P2; ++i) {
        const int v = tid + NT * i, row = v / (N1 / 8), c = v % (N1 / 8);
        *reinterpret_cast<u32x4_t*>(s_w2 + row * ROW2 + c * 8) = w2r[i];
    }
    __syncthreads();

    for (int64_t row0 = wg_first + wave * 32; row0 < wg_end; row0 += NT / 2) {       // wave-uniform
        const int64_t my_row = row0 + r;
        uint16_t* sh = s_h[wave];
        const bool more = row0 + NT / 2 < wg_end;
        // ---- layer-1 B fragments: lane (b, hf) holds h[b][16 s + 8 hf .. + 7] for every k-step s
        // (a wave's LDS operations execute in program order: the fragment reads of one half precede the writes of the next -- no
        // barrier, the region is this wave's own)
        bf16x8_t bx[K1 / 16];
        const uint16_t* hr = sh + r * ROWS + 8 * hf;
        sfor<0, 2>([&](auto HALF) {
            constexpr int half = decltype(HALF)::value;
            sfor<0, KH / 16>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const int pc = 64 * i + lane, prow = pc >> 4, pcol = pc & 15;
                *reinterpret_cast<u32x4_t*>(sh + prow * ROWS + pcol * 8) = pieces[half][i];
            });
            if (more) request(row0 + NT / 2, HALF);
#pragma unroll
            for (int s = 0; s < KH / 16; ++s) bx[half * (KH / 16) + s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(hr + 16 * s));   // the type the writes use: the two halves alias
        });

        // ---- layer 1: four 32-feature tiles in two passes of two (at two waves per SIMD a wave has 256 registers: four accumulator
        // tiles beside the row fragments and the prefetched next rows spill), k outer: the accumulator chains of a pass are
        // independent, and the A fragments of k-step s + 1 are requested from LDS before the MFMAs of k-step s are issued (left
        // to the compiler every ds_read_b128 landed in the registers its MFMA then consumed: read, wait, MFMA).  The base address
        // is laundered once per tile of rows: the fragment reads must stay in this loop.
        // bias + ReLU + bf16: elements 8 q .. 8 q + 7 of tile t are the B fragment of layer-2 k-step 2 t + q
        bf16x8_t b2f[N1 / 16];
        const uint16_t* wr1 = s_w1 + r * ROW1 + 8 * hf;
        asm volatile("" : "+v"(wr1));
        sfor<0, 2>([&](auto NP) {
            constexpr int t0 = 2 * decltype(NP)::value;
            f32x16_t acc1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)                  // the accumulators start from the biases (one add per element less in the epilogue)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 bb = *reinterpret_cast<const float4*>(s_b1 + 32 * (t0 + t) + 8 * j + 4 * hf);
                    acc1[t][4 * j] = bb.x; acc1[t][4 * j + 1] = bb.y; acc1[t][4 * j + 2] = bb.z; acc1[t][4 * j + 3] = bb.w;
                }
            u32x4_t an[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) an[0][t] = *reinterpret_cast<const u32x4_t*>(wr1 + 32 * (t0 + t) * ROW1);
#pragma unroll
            for (int s = 0; s < K1 / 16; ++s) {
                if (s + 1 < K1 / 16) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) an[(s + 1) & 1][t] = *reinterpret_cast<const u32x4_t*>(wr1 + 32 * (t0 + t) * ROW1 + 16 * (s + 1));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, an[s & 1][t]), bx[s], acc1[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bf16x8_t f;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * q + jj;                                      // e >> 2
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc1[t][4 * j + i];
                            f[4 * jj + i] = static_cast<__bf16>(v > 0.0f ? v : 0.0f);
                        }
                    }
                    b2f[2 * (t0 + t) + q] = f;
                }
            }
        });

        // ---- layer 2: two 32-feature tiles (same arrangement), stored as 8-byte pieces (4 adjacent features of this lane's row)
        f32x16_t acc2[N2 / 32];
#pragma unroll
        for (int t = 0; t < N2 / 32; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bb = *reinterpret_cast<const float4*>(s_b2 + 32 * t + 8 * j + 4 * hf);
                acc2[t][4 * j] = bb.x; acc2[t][4 * j + 1] = bb.y; acc2[t][4 * j + 2] = bb.z; acc2[t][4 * j + 3] = bb.w;
            }
        const uint16_t* wr2 = s_w2 + r * ROW2 + 8 * hf;
        asm volatile("" : "+v"(wr2));
        u32x4_t cn[2][N2 / 32];
#pragma unroll
        for (int t = 0; t < N2 / 32; ++t) cn[0][t] = *reinterpret_cast<const u32x4_t*>(wr2 + 32 * t * ROW2);
#pragma unroll
        for (int s = 0; s < N1 / 16; ++s) {
            if (s + 1 < N1 / 16) {
#pragma unroll
                for (int t = 0; t < N2 / 32; ++t) cn[(s + 1) & 1][t] = *reinterpret_cast<const u32x4_t*>(wr2 + 32 * t * ROW2 + 16 * (s + 1));
            }
#pragma unroll
            for (int t = 0; t < N2 / 32; ++t)
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, cn[s & 1][t]), b2f[s], acc2[t], 0, 0, 0);
        }
        bf16x8_t b3f[N2 / 16];                           // HEADS: layer 2's outputs as the head layer's B fragments
#pragma unroll
        for (int t = 0; t < N2 / 32; ++t) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8_t f;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * q + jj;
                    bf16x4_t o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc2[t][4 * j + i];
                        o[i] = static_cast<__bf16>(v > 0.0f ? v : 0.0f);
                        f[4 * jj + i] = o[i];
                    }
                    if constexpr (!HEADS) {
                        if (my_row < B) *reinterpret_cast<uint2*>(lat + my_row * N2 + 32 * t + 8 * j + 4 * hf) = __builtin_bit_cast(uint2, o);
                    }
                }
                b3f[2 * t + q] = f;
            }
        }
        if constexpr (HEADS) {
            f32x16_t acc3;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc3[e] = 0.0f;
            const uint16_t* wrh = s_hw + r * ROWH + 8 * hf;
#pragma unroll
            for (int s3 = 0; s3 < N2 / 16; ++s3)
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrh + 16 * s3)), b3f[s3], acc3, 0, 0, 0);
            // output n of this lane's row sits in element n of the lanes with hf = 0 (n = (e & 3) + 8 (e >> 2) + 4 hf)
            const float hm[4] = { acc3[0], acc3[1], acc3[2], acc3[3] };
            if (hf == 0 && my_row < B) {
                if (trunk == 1) {
                    hd.value[my_row] = hm[0] + s_hb[0];
                } else {
                    float z[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                    if (!hd.deterministic) {                                           // as policy_heads_kernel: same key, same Box-Muller
                        uint32_t rn[4];
                        philox4(hd.seed, uint32_t(my_row), uint32_t(my_row >> 32), hd.step ? *hd.step : 0u, 0x51u, rn);
                        const float u0 = (float(rn[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float(rn[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        const float u2 = (float(rn[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float(rn[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
                        z[0] = ra * __cosf(6.283185307f * u1); z[1] = ra * __sinf(6.283185307f * u1);
                        z[2] = rb * __cosf(6.283185307f * u3); z[3] = rb * __sinf(6.283185307f * u3);
                    }
                    float a[4], lp = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float ls = hd.log_std[k];
                        a[k] = hm[k] + s_hb[k] + __expf(ls) * z[k];
                        lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
                    }
                    reinterpret_cast<float4*>(hd.actions)[my_row] = make_float4(a[0], a[1], a[2], a[3]);
                    hd.logp[my_row] = lp;
                }
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
    const int64_t blocks = (B + NT / 2 - 1) / (NT / 2);                      // 32 rows per wave and pass
    const int64_t gx = blocks < TRUNK_WGS ? blocks : TRUNK_WGS;
    const int64_t rows_per_wg = ((blocks + gx - 1) / gx) * (NT / 2);
    hipLaunchKernelGGL(policy_trunk_kernel<false>, dim3(unsigned(gx), 2), dim3(NT), 0, (hipStream_t)stream, (const uint16_t*)h_pi,
                       (const uint16_t*)h_vf, (const uint16_t*)W1, b1, (const uint16_t*)W2p, b2, (uint16_t*)lat_pi, (uint16_t*)lat_vf, B,
                       rows_per_wg, HeadArgs{});
    return int(hipGetLastError());
}

extern "C" int fdyn_policy_trunks_heads(const void* h_pi, const void* h_vf, const void* W1, const float* b1, const void* W2p, const float* b2,
                                        const void* Wa, const void* ba, const void* wv, const void* bv, const float* log_std, uint64_t seed,
                                        const uint32_t* step, int deterministic, float* actions, float* logp, float* value, int64_t B,
                                        void* stream)
{
    if (B < 0) return FDYN_ERR_BAD_SIZE;
    if (!h_pi || !h_vf || !W1 || !b1 || !W2p || !b2 || !Wa || !ba || !wv || !bv || !log_std || !actions || !logp || !value) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const int64_t blocks = (B + NT / 2 - 1) / (NT / 2);                      // 32 rows per wave and pass
    const int64_t gx = blocks < TRUNK_WGS ? blocks : TRUNK_WGS;
    const int64_t rows_per_wg = ((blocks + gx - 1) / gx) * (NT / 2);
    const HeadArgs hd = { (const uint16_t*)Wa, (const uint16_t*)ba, (const uint16_t*)wv, (const uint16_t*)bv, log_std, seed, step, deterministic,
                          actions, logp, value };
    hipLaunchKernelGGL(policy_trunk_kernel<true>, dim3(unsigned(gx), 2), dim3(NT), 0, (hipStream_t)stream, (const uint16_t*)h_pi,
                       (const uint16_t*)h_vf, (const uint16_t*)W1, b1, (const uint16_t*)W2p, b2, (uint16_t*)nullptr, (uint16_t*)nullptr, B,
                       rows_per_wg, hd);
    return int(hipGetLastError());
}
