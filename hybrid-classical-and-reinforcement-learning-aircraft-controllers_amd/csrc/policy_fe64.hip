// policy_fe64.hip -- the policy's whole features extractor as ONE kernel:
//
//   obs [B][18] fp32 -> Linear(18,128)+ReLU -> zero-state LSTM layer 128->256 -> zero-state LSTM layer 256->256
//                    -> Linear(256,128)+ReLU -> feats [B][128] bf16
//
// (learned_controllers/networks/lstm_policy.py:13-97: the reference runs nn.LSTM on a length-1 sequence without carried state,
// so each layer is h = sigmoid(o) tanh(sigmoid(i) tanh(g)), gates = x W_ih^T + b_ih + b_hh -- policy.py explains.)
//
// As four launches (hipBLASLt GEMM, two lstm_cell_mfma_dma_kernel, hipBLASLt GEMM) this chain moved 190 MB through HBM per
// 65 536-row step and took 107 us, 79 us of it in the two cells at 12-20 % of the matrix pipe.  Here a workgroup carries its
// 256 rows through all four layers in registers: 22 MB of HBM traffic (obs in, feats out), the rest is MFMA and VALU.
//
// Shape (the one-wave-per-SIMD scheme of lstm_mfma64.hip): 4 waves x 64 rows, 512 registers per lane.
//   * SWAPPED operand roles: the weights are the MFMA A operand (read from LDS), the activations the B operand (registers), so
//     an output tile is D[n][b]: lane (b, hf) holds, for ITS OWN batch row b, 16 hidden units n = (e & 3) + 8 (e >> 2) + 4 hf of
//     the 32-unit slice.  Registers 8 hq .. 8 hq + 7 of that tile, rounded to bf16 and packed, ARE the next layer's B fragment
//     for k-step 2 slice + hq -- no transposition, no LDS, no HBM (cdna_hip_programming.md, "an accumulator tile as the next
//     MFMA's operand").  The price is a fixed permutation of k inside every block of 16, which the host folds into the
//     weight image once per weight refresh (policy.py: pack_fe_weights, kperm below).
//   * activations between layers live in accumulator registers that only inline assembly touches: R1 = a[0:127] (h of layer
//     1), R2 = a[128:255] (embedding output, later h of layer 2); the compiler owns the 256 architectural registers.  It
//     has NO accumulator register to fall back on here: __graft_entry__.build() checks that it did not take one.
//   * weights: the host lays every chunk out exactly as its LDS image (rows padded by 16 B), chunks in the order they are
//     used; a chunk is fetched by LDS-DMA as 1 KB pieces, two LDS buffers, chunk q + 1 requested while chunk q multiplies.
//   * per 32-unit slice two units: (i, g) then (o); the point-wise work of a unit runs under the next unit's MFMAs, dealt
//     out in micro-stages (one operation on four elements) after each MFMA, fenced so the compiler keeps the placement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/fdyn.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int H = 256, EMB = 128, FEAT = 128, OBS = 18;
constexpr int R1 = 0, R2 = 128;                     // accumulator-register regions (see above)
constexpr float L2E = 1.4426950408889634f;

// ---- weight image (host: policy.py pack_fe_weights; keep the two in step) -------------------------------------------------
// chunk = rows x (2 K + 16) bytes, padded to a multiple of 1 KB (one DMA piece)
constexpr int rowb(int K) { return 2 * K + 16; }
constexpr int pieces(int rows, int K) { return (rows * rowb(K) + 1023) / 1024; }
constexpr int NP_EMB = pieces(128, 32);             // 10
constexpr int NP_A_IG = pieces(64, 128), NP_A_O = pieces(32, 128);      // 17, 9
constexpr int NP_B_IG = pieces(64, 256), NP_B_O = pieces(32, 256);      // 33, 17
constexpr int NP_C = pieces(32, 256);               // 17
constexpr int OFF_A = NP_EMB, OFF_B = OFF_A + 8 * (NP_A_IG + NP_A_O), OFF_C = OFF_B + 8 * (NP_B_IG + NP_B_O);   // in pieces
constexpr int IMG_PIECES = OFF_C + 4 * NP_C;
constexpr int BUFB = NP_B_IG * 1024;                // bytes of one LDS weight buffer (the largest chunk)
constexpr int NPW_MAX = (NP_B_IG + 3) / 4;          // DMA pieces one wave issues for the largest chunk (9)

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int V> using IC = std::integral_constant<int, V>;

// 1 KB from the weight image (scalar address, lane * 16 added by the hardware path below) to LDS byte offset lds_off
__device__ __forceinline__ void dma_piece(const void* src, uint32_t lane16, uint32_t lds_off)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_off), "v"(lane16), "s"(src) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// acc += W(fragment in architectural registers) x X(a[LO:LO+3]); `zero` starts a chain
template <int LO> __device__ __forceinline__ void mfma_a(f32x16_t& acc, const bf16x8_t& w)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma_a_first(f32x16_t& acc, const bf16x8_t& w)
{
    asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma_a0(f32x16_t& acc, const bf16x8_t& w)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
__device__ __forceinline__ void mfma_v(f32x16_t& acc, const bf16x8_t& w, const bf16x8_t& x)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
__device__ __forceinline__ void mfma_v0(f32x16_t& acc, const bf16x8_t& w, const bf16x8_t& x)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
}
template <int A> __device__ __forceinline__ void wr_a(uint32_t v) { asm volatile("v_accvgpr_write_b32 a[%1], %0" :: "v"(v), "n"(A)); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t p = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// the glue between the previous env step and this policy step (what fdyn_episode_flags does in a launch of its own), done by
// the first kernel of the step: thread = row
struct FlagArgs { const uint8_t* term; const uint8_t* trunc; float* episode_start; float* keep; int32_t* counter; };

__global__ void __launch_bounds__(256, 1)
policy_fe64_kernel(const float* __restrict__ obs /*[B][18]*/, const uint8_t* __restrict__ wimg /*[IMG_PIECES][1024]*/,
                   const float* __restrict__ bias /*[128 + 1024 + 1024 + 128]: embedding, layer 1, layer 2, projection*/,
                   uint16_t* __restrict__ feats /*[B][128] bf16*/, FlagArgs fl)
{
    constexpr int OROW = 2 * FEAT + 16;                 // padded row of the output staging tile (bytes)
    __shared__ __attribute__((aligned(16))) uint8_t s_w[2 * BUFB];
    __shared__ __attribute__((aligned(16))) float s_b[EMB + 4 * H + 4 * H + FEAT];      // pre-scaled biases
    __shared__ __attribute__((aligned(16))) uint8_t s_out[4][64 * OROW];                 // per wave: 64 rows of feats on their way out
    constexpr int SB_E = 0, SB_1 = EMB, SB_2 = EMB + 4 * H, SB_P = EMB + 8 * H;

    asm volatile("" ::: "a0", "a255");                  // a[0:255] belong to the inline assembly below

    const int tid = threadIdx.x, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int64_t urow0 = int64_t(blockIdx.x) * 256 + uwave * 64;
    const uint32_t lane16 = uint32_t(lane * 16);
    const uint32_t lds_w = uint32_t(reinterpret_cast<uintptr_t>(&s_w[0]));

    // chunk request: pieces uwave, uwave + 4, ... of chunk [off, off + np) -> LDS buffer buf; SLOT-th piece of this wave
    auto req_piece = [&](int slot, int off, int np, int buf) {
        const int p = uwave + 4 * slot;
        if (p < np) dma_piece(wimg + (int64_t(off) + p) * 1024, lane16, lds_w + uint32_t(buf * BUFB + p * 1024));
    };
    auto req_all = [&](int off, int np, int buf) {
#pragma unroll
        for (int s = 0; s < NPW_MAX; ++s) req_piece(s, off, np, buf);
    };
    req_all(0, NP_EMB, 0);
    req_all(OFF_A, NP_A_IG, 1);

    // ---- every load of the prologue is issued before the first of them is waited for (round 3: as a rolled load -> scale -> LDS
    // loop the nine bias words of a thread were nine dependent L2 round trips -- most of the 12 k cycles in front of the embedding,
    // profiles/r03_fe64_ns7.log -- and the flag bytes a tenth): flags, biases, observations; then their consumers.
    const int64_t frow = int64_t(blockIdx.x) * 256 + tid;
    uint8_t fterm = 0, ftrunc = 0;
    if (fl.term) { fterm = fl.term[frow]; ftrunc = fl.trunc[frow]; }
    constexpr int NBV = (EMB + 8 * H + FEAT) / 256;
    static_assert((EMB + 8 * H + FEAT) % 256 == 0, "whole bias words per thread");
    float bvals[NBV];
#pragma unroll
    for (int j = 0; j < NBV; ++j) bvals[j] = bias[tid + 256 * j];
    // ---- observation fragments: lane (b, hf) holds k = 16 ks + 8 hf .. + 7 of its row for ks = 0, 1 (k >= 18 is zero)
    bf16x8_t xo[2][2];
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* orow = obs + (urow0 + 32 * t + r) * OBS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int k = 16 * ks + 8 * hf + p;           // hf is per lane: select, do not branch
                    const float x = orow[k < OBS ? k : 0];
                    v[p] = k < OBS ? x : 0.0f;
                }
                u32x4_t u = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                xo[t][ks] = __builtin_bit_cast(bf16x8_t, u);
            }
        }
    }
    if (fl.term) {                                      // episode_start / keep of this step's rows, the action-noise counter
        const bool done = (fterm | ftrunc) != 0;
        if (fl.episode_start) fl.episode_start[frow] = done ? 1.0f : 0.0f;
        if (fl.keep) fl.keep[frow] = done ? 0.0f : 1.0f;
        if (fl.counter && blockIdx.x == 0 && tid == 0) fl.counter[0] += 1;
    }
    // ---- biases -> LDS, pre-scaled like their weight rows (policy.py: pack_fe_weights): i, o: -log2 e; g: -2 log2 e; Linear layers: 1
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
        const int i = tid + 256 * j;
        float sc = 1.0f;
        if (i >= SB_1 && i < SB_P) { const int gate = ((i - SB_1) % (4 * H)) / H; sc = gate == 2 ? -2.0f * L2E : -L2E; }
        s_b[i] = bvals[j] * sc;
    }
    wait_vm0();
    __syncthreads();

    // weight fragment (A operand): lane (n = r, hf) reads 16 bytes of row `row0 + r`, k-step ks
    auto wfrag = [&](int buf, int rowbytes, int row0, int ks) {
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(&s_w[buf * BUFB + (row0 + r) * rowbytes + ks * 32 + hf * 16]));
    };

    // =========================================== embedding: Linear(18 -> 128) + ReLU -> R2 (8 k-steps of layer 1) ===========
    // small (16 MFMAs): straight code, no pipelining.  Requests chunk (layer 1, slice 0, o) afterwards.
    sfor<0, 4>([&](auto SL) {
        constexpr int sl = decltype(SL)::value;
        f32x16_t acc[2];
        const bf16x8_t w0 = wfrag(0, rowb(32), 32 * sl, 0), w1 = wfrag(0, rowb(32), 32 * sl, 1);
        mfma_v0(acc[0], w0, xo[0][0]); mfma_v0(acc[1], w0, xo[1][0]);
        mfma_v(acc[0], w1, xo[0][1]); mfma_v(acc[1], w1, xo[1][1]);
        asm volatile("s_nop 15\n\ts_nop 3");
        sfor<0, 2>([&](auto T) {
            constexpr int t = decltype(T)::value;
            sfor<0, 4>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(&s_b[SB_E + 32 * sl + 8 * q + 4 * hf]);
                float y[4];
                sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; y[i] = __builtin_fmaxf(acc[t][4 * q + i] + b4[i], 0.0f); });
                // registers 8 hq .. 8 hq + 7 -> fragment (k-step 2 sl + hq), dwords 2 (q & 1), 2 (q & 1) + 1
                wr_a<R2 + 4 * (8 * t + 2 * sl + (q >> 1)) + 2 * (q & 1)>(pack2(y[0], y[1]));
                wr_a<R2 + 4 * (8 * t + 2 * sl + (q >> 1)) + 2 * (q & 1) + 1>(pack2(y[2], y[3]));
            });
        });
    });
    __syncthreads();                                    // everyone is done with buffer 0
    req_all(OFF_A + NP_A_IG, NP_A_O, 0);

    // =========================================== a zero-state LSTM layer ====================================================
    // input slab a[IN .. IN + 8 KS) (fragment (t, ks) at IN + 4 (KS t + ks)), output slab a[OUT ..) with 16 k-steps per tile.
    f32x16_t aA[2][2], aB[2];                           // (i, g) x tiles; o x tiles
    float ig[2][16];
    uint32_t hp[2][8];                                  // packed h of the pending slice: tile t, dwords 4 hq + j of fragment hq

    // one unit: NG gates x KS k-steps x 2 tiles MFMAs from LDS buffer `buf`; after MFMA m the next chunk's DMA slot (while
    // any are left) and the micro-stages [m U / M, (m + 1) U / M) of `micro`
    // `pre` runs at the top of the unit, behind the first weight-fragment reads: it loads the unit's accumulators with the gates'
    // biases (round 3, third form: the weight image and the biases arrive pre-scaled by the factor their gate's exponent needs --
    // policy.py: pack_fe_weights -- and the bias rides in the accumulator, so the epilogue starts at ex2(acc): two FMAs per element
    // less in the (i, g) epilogue, one in the h epilogue, of 9 -- these layers are bound by the vector instructions of the
    // epilogue, not by the MFMAs: scratch/ubench/mfma_shadow.hip)
    auto unit = [&](auto ks_c, auto ng_c, auto in_c, int buf, auto& acc_of, auto nmicro_c, auto&& micro, int noff, int nnp, int nbuf, auto&& pre) {
        constexpr int KS = decltype(ks_c)::value, NG = decltype(ng_c)::value, IN = decltype(in_c)::value;
        constexpr int M = KS * NG * 2, U = decltype(nmicro_c)::value;
        constexpr int EVERY = M / NPW_MAX >= 3 ? 3 : (M / NPW_MAX >= 1 ? M / NPW_MAX : 1);
        constexpr int RB = rowb(16 * KS);
        bf16x8_t w[2][NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) w[0][g] = wfrag(buf, RB, 32 * g, 0);
        pre();
        FENCE();
        sfor<0, KS>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            sfor<0, NG * 2>([&](auto MM) {
                constexpr int mm = decltype(MM)::value, g = mm >> 1, t = mm & 1, m = ks * NG * 2 + mm;
                if constexpr (ks + 1 < KS && t == 0) w[(ks + 1) & 1][g] = wfrag(buf, RB, 32 * g, ks + 1);
                // onto the biases `pre` left there; they reach half of the accumulators through v_mov, placed by the compiler
                // right in front of this asm, which it cannot see is an MFMA: the VALU write -> MFMA SrcC distance rides inside
                if constexpr (ks == 0) mfma_a_first<IN + 4 * (KS * t + ks)>(acc_of(IC<g>{}, IC<t>{}), w[ks & 1][g]);
                else mfma_a<IN + 4 * (KS * t + ks)>(acc_of(IC<g>{}, IC<t>{}), w[ks & 1][g]);
                if constexpr (m % EVERY == 0 && m / EVERY < NPW_MAX) req_piece(m / EVERY, noff, nnp, nbuf);
                sfor<(m * U) / M, ((m + 1) * U) / M>([&](auto UU) { micro(UU); });
                FENCE();
            });
        });
        asm volatile("s_nop 15\n\ts_nop 3");            // MFMA result -> VALU read distance (the compiler cannot see into the asm)
    };

    // micro-stage order: groups in pairs, the two groups of a pair alternate (dependent operations sit eight instructions apart)
    constexpr int NS = 7;                               // micro-stages per group in both epilogues (round 2: 11, round 3: 9, then 7)
    constexpr int NMICRO = 8 * NS;
    constexpr float CG = -2.0f * L2E;                   // the cell value leaves the (i, g) epilogue as CG c: the h epilogue's exponent
    struct GS { float x[4], y[4], z[4]; };

    auto layer = [&](auto ks_c, auto in_c, auto out_c, int sb, int off0, int next_off, int next_np) {
        constexpr int KS = decltype(ks_c)::value, IN = decltype(in_c)::value, OUT = decltype(out_c)::value;
        constexpr int NP_IG = pieces(64, 16 * KS), NP_O = pieces(32, 16 * KS);
        GS gs[8];
        int psl = 0;                                    // the slice whose h epilogue is pending
        // (i, g) epilogue of slice sl, micro-stage U: with I = 2^acc_i = e^-i and G = 2^acc_g = e^-2g (the accumulators hold the
        // scaled pre-activations), ig = CG sigmoid(i) tanh(g) = CG (1 - G) / ((1 + G)(1 + I)): the constant rides in (1 + G) / CG
        auto ig_micro = [&](int sl, auto UU) {
            constexpr int u = decltype(UU)::value, pr = u / (2 * NS), st = (u % (2 * NS)) / 2, gq = 2 * pr + (u & 1);
            constexpr int t = gq >> 2, q = gq & 3;
            GS& s = gs[gq];
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = ex2(aA[0][t][4 * q + i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = __builtin_amdgcn_fmed3f(aA[1][t][4 * q + i], -40.0f, 40.0f); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = ex2(s.y[i]); });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.y[i], 1.0f / CG, 1.0f / CG); });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.z[i], s.x[i], s.z[i]); });
            } else if constexpr (st == 5) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = rcp(s.z[i]); });
            } else { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; ig[t][4 * q + i] = __builtin_fmaf(-s.y[i], s.z[i], s.z[i]); }); }
        };
        // o epilogue of the pending slice: h = sigmoid(o) tanh(c) = (1 - E) / ((1 + E)(1 + O)), O = 2^acc_o = e^-o, E = 2^(CG c) = e^-2c
        // (c in (-1, 1): no clamp)
        auto h_micro = [&](auto UU) {
            constexpr int u = decltype(UU)::value, pr = u / (2 * NS), st = (u % (2 * NS)) / 2, gq = 2 * pr + (u & 1);
            constexpr int t = gq >> 2, q = gq & 3;
            GS& s = gs[gq];
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = ex2(aB[t][4 * q + i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = ex2(ig[t][4 * q + i]); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = 1.0f + s.y[i]; });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.z[i], s.x[i], s.z[i]); });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = rcp(s.z[i]); });
            } else if constexpr (st == 5) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(-s.y[i], s.z[i], s.z[i]); });
            } else { hp[t][2 * q] = pack2(s.z[0], s.z[1]); hp[t][2 * q + 1] = pack2(s.z[2], s.z[3]); }
        };
        // the pending slice's packed h -> output slab (the register index must be an immediate: one arm per slice)
        auto commit = [&]() {
            auto wr = [&](auto PS) {
                constexpr int ps = decltype(PS)::value;
                sfor<0, 2>([&](auto T) { sfor<0, 8>([&](auto J) {
                    constexpr int t = decltype(T)::value, j = decltype(J)::value;
                    wr_a<OUT + 4 * (16 * t + 2 * ps + (j >> 2)) + (j & 3)>(hp[t][j]);
                }); });
            };
            switch (psl) {
                case 0: wr(IC<0>{}); break; case 1: wr(IC<1>{}); break; case 2: wr(IC<2>{}); break; case 3: wr(IC<3>{}); break;
                case 4: wr(IC<4>{}); break; case 5: wr(IC<5>{}); break; case 6: wr(IC<6>{}); break; default: wr(IC<7>{}); break;
            }
        };
        auto accA = [&](auto G, auto T) -> f32x16_t& { return aA[decltype(G)::value][decltype(T)::value]; };
        auto accB = [&](auto, auto T) -> f32x16_t& { return aB[decltype(T)::value]; };

        for (int sl = 0; sl < 8; ++sl) {
            const int off_ig = off0 + sl * (NP_IG + NP_O), off_o = off_ig + NP_IG;
            // P0: (i, g) of slice sl from buffer 1 -> aA ; under it the o epilogue of slice sl - 1 ; request (sl, o) -> buffer 0
            // the accumulators of the unit about to run <- its gates' (pre-scaled) biases: element 4 q + i of a tile is unit
            // 32 sl + 8 q + 4 hf + i, the same for both row tiles
            auto pre_ig = [&]() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t bi = *reinterpret_cast<const f32x4_t*>(&s_b[sb + 32 * sl + 8 * q + 4 * hf]);
                    const f32x4_t bg = *reinterpret_cast<const f32x4_t*>(&s_b[sb + 2 * H + 32 * sl + 8 * q + 4 * hf]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { aA[0][0][4 * q + i] = bi[i]; aA[0][1][4 * q + i] = bi[i]; aA[1][0][4 * q + i] = bg[i]; aA[1][1][4 * q + i] = bg[i]; }
                }
            };
            auto pre_o = [&]() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t bo = *reinterpret_cast<const f32x4_t*>(&s_b[sb + 3 * H + 32 * sl + 8 * q + 4 * hf]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { aB[0][4 * q + i] = bo[i]; aB[1][4 * q + i] = bo[i]; }
                }
            };
            if (sl == 0) unit(ks_c, IC<2>{}, in_c, 1, accA, IC<0>{}, [&](auto) {}, off_o, 0, 0, pre_ig);       // its o chunk was requested before the layer
            else unit(ks_c, IC<2>{}, in_c, 1, accA, IC<NMICRO>{}, h_micro, off_o, NP_O, 0, pre_ig);
            if (sl > 0) commit();
            wait_vm0();
            __syncthreads();
            // P1: o of slice sl from buffer 0 -> aB ; under it the (i, g) epilogue of slice sl ; request the next (i, g) chunk
            // (past the layer: the next layer's first chunk) -> buffer 1
            const int noff = sl < 7 ? off_ig + NP_IG + NP_O : next_off, nnp = sl < 7 ? NP_IG : next_np;
            unit(ks_c, IC<1>{}, in_c, 0, accB, IC<NMICRO>{}, [&](auto UU) { ig_micro(sl, UU); }, noff, nnp, 1, pre_o);
            psl = sl;
            wait_vm0();
            __syncthreads();
        }
        // drain: the last slice's o epilogue has nothing to hide under
        sfor<0, NMICRO>([&](auto UU) { h_micro(UU); });
        commit();
    };
    // layer 1: K = 128 from R2 -> R1 ; its first (i, g) chunk was requested at the top, its first o chunk after the embedding
    layer(IC<8>{}, IC<R2>{}, IC<R1>{}, SB_1, OFF_A, OFF_B, NP_B_IG);
    // layer 2: K = 256 from R1 -> R2
    __syncthreads();
    req_all(OFF_B + NP_B_IG, NP_B_O, 0);
    layer(IC<16>{}, IC<R1>{}, IC<R2>{}, SB_2, OFF_B, OFF_C, NP_C);

    // =========================================== projection: Linear(256 -> 128) + ReLU -> feats =============================
    // chunk s (32 rows) in buffer (s + 1) & 1 (slice 0 was requested into buffer 1 by layer 2's last unit)
    for (int sl = 0; sl < 4; ++sl) {
        const int buf = (sl + 1) & 1;
        if (sl < 3) req_all(OFF_C + (sl + 1) * NP_C, NP_C, buf ^ 1);
        f32x16_t acc[2];
        bf16x8_t w[2];
        w[0] = wfrag(buf, rowb(256), 0, 0);
        sfor<0, 16>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            if constexpr (ks + 1 < 16) w[(ks + 1) & 1] = wfrag(buf, rowb(256), 0, ks + 1);
            if constexpr (ks == 0) { mfma_a0<R2 + 4 * ks>(acc[0], w[0]); mfma_a0<R2 + 4 * (16 + ks)>(acc[1], w[0]); }
            else { mfma_a<R2 + 4 * ks>(acc[0], w[ks & 1]); mfma_a<R2 + 4 * (16 + ks)>(acc[1], w[ks & 1]); }
            FENCE();
        });
        asm volatile("s_nop 15\n\ts_nop 3");
        sfor<0, 2>([&](auto T) {
            constexpr int t = decltype(T)::value;
            sfor<0, 4>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(&s_b[SB_P + 32 * sl + 8 * q + 4 * hf]);
                float y[4];
                sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; y[i] = __builtin_fmaxf(acc[t][4 * q + i] + b4[i], 0.0f); });
                uint2 pk = {pack2(y[0], y[1]), pack2(y[2], y[3])};     // features 32 sl + 8 q + 4 hf .. + 3 of row 32 t + r
                *reinterpret_cast<uint2*>(&s_out[uwave][(32 * t + r) * OROW + (32 * sl + 8 * q + 4 * hf) * 2]) = pk;
            });
        });
        wait_vm0();
        __syncthreads();
    }
    // ---- feats: whole 256-byte rows from the staging tile (16 lanes per row, 4 rows per instruction)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = 4 * it + (lane >> 4), cb = (lane & 15) * 16;
        const uint4 v = *reinterpret_cast<const uint4*>(&s_out[uwave][row * OROW + cb]);
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(feats + (urow0 + row) * FEAT) + cb) = v;
    }
}

}  // namespace

extern "C" int fdyn_policy_features_image_bytes(void) { return IMG_PIECES * 1024; }

extern "C" int fdyn_policy_features(const float* obs, const void* weight_image, const float* bias, void* feats, int64_t B, void* stream)
{
    if (!obs || !weight_image || !bias || !feats) return FDYN_ERR_NULL;
    if (B <= 0 || B % 256) return FDYN_ERR_BAD_SIZE;
    hipLaunchKernelGGL(policy_fe64_kernel, dim3(unsigned(B / 256)), dim3(256), 0, (hipStream_t)stream,
                       obs, (const uint8_t*)weight_image, bias, (uint16_t*)feats, FlagArgs{});
    return int(hipGetLastError());
}

extern "C" int fdyn_policy_features_flags(const float* obs, const void* weight_image, const float* bias, void* feats,
                                          const uint8_t* terminated, const uint8_t* truncated, float* episode_start, float* keep,
                                          int32_t* counter, int64_t B, void* stream)
{
    if (!obs || !weight_image || !bias || !feats || !terminated || !truncated) return FDYN_ERR_NULL;
    if (B <= 0 || B % 256) return FDYN_ERR_BAD_SIZE;
    hipLaunchKernelGGL(policy_fe64_kernel, dim3(unsigned(B / 256)), dim3(256), 0, (hipStream_t)stream,
                       obs, (const uint8_t*)weight_image, bias, (uint16_t*)feats, FlagArgs{terminated, truncated, episode_start, keep, counter});
    return int(hipGetLastError());
}
