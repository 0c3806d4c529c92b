// policy_rc64.hip -- the policy's TWO recurrent cells (sb3_contrib MlpLstmPolicy's actor and critic nn.LSTM(128, 256),
// learned_controllers/networks/lstm_policy.py:107-136 + train_rate.py:128-147) as ONE launch, lane = batch row.
//
//   feats [B][128] bf16, h [2][B][256] bf16, c [2][B][256] fp32, keep [B]  ->  h', c' (IN PLACE when out == in)
//   gates = [feats | keep h] W^T + b ; c' = sigmoid(f) keep c + sigmoid(i) tanh(g) ; h' = sigmoid(o) tanh(c')
//
// What was there: lstm_mfma64.hip once per cell (activations = MFMA A operand, weights = B from LDS).  Its output tile is
// D[b][n] -- a lane owns ONE hidden unit of 16 rows -- so c_prev / c' / h' moved as 4- and 2-byte words, 64-128 contiguous bytes
// per row at a 1 KB row stride: 218 MB per cell at 3.45 TB/s, under both roofs (32.6 % of the matrix pipe), and the two cells
// read the same features in two launches against two ping-pong state sets (400 MB of state: nothing survives in the 256 MB
// Infinity Cache from one step to the next).
//
// Here (the operand roles of policy_fe64.hip):
//   * WEIGHTS are the MFMA A operand (LDS), ACTIVATIONS the B operand (registers): the output tile is D[n][b], lane (b, hf)
//     holds 16 hidden units n = (e & 3) + 8 (e >> 2) + 4 hf of ITS OWN batch row.  Everything a lane loads or stores is its own
//     row's, so the state can be updated IN PLACE (201 MB of state instead of 400) and laid out for the lanes:
//       c [B/64][slice 8][tile 2][j 4][lane 64][4 fp32]   one 16-byte load and one 16-byte store per lane and group, every wave
//                                                          instruction a contiguous 1 KB
//       h [B/64][tile 2][k-step 16][lane 64][8 bf16]       the packed output registers 8 hq .. 8 hq + 7 of slice sl ARE the B
//                                                          fragment of k-step 2 sl + hq of the next step (and of the trunks):
//                                                          stored as they stand, loaded back with one 16-byte load per fragment
//     (policy.py: rc_pack_h / rc_pack_c convert to and from the [B][256] row-major form the BPTT update reads; the fixed order
//     of k inside every block of 16, _KPERM16, is folded into the weight image by pack_rc_weights.)
//   * both cells in one launch: the feature fragments are loaded once; the critic's chunks follow the actor's in one stream.
//   * one wave per SIMD, 512 registers per lane, split by hand: a[0:63] feature fragments, a[64:191] h fragments of the cell
//     being computed, a[192:223] the slice's c_prev on its way in (requested a whole slice = four units ahead);
//     the compiler owns the 256 architectural registers (two accumulator sets 64, the slice's running values 32, fragments).
//   * a unit = ONE gate x 32 hidden units x all K = 384 columns (48 MFMAs, 24.5 KB of weights); four LDS buffers, chunk u + 3
//     requested by LDS-DMA at the top of unit u; one barrier per unit.  The point-wise work of unit u runs under the MFMAs of
//     unit u + 1, dealt out in micro-stages (one operation on four elements) after each MFMA, fenced.
//   * vector-memory waits are hand-counted (every global access after the prologue is inline assembly or a plain store): at
//     the end of unit u only what is older than chunk u + 2's request has to have retired.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/fdyn.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int H = 256, KX = 128, K = KX + H;
constexpr int KSTEPS = K / 16, XSTEPS = KX / 16, HSTEPS = H / 16;
constexpr int NSLICE = H / 32;
constexpr int NUNITS = 2 * NSLICE * 4;                // global unit (= chunk) index: ((cell * 8 + slice) * 4 + gate position i, g, f, o)
constexpr int ROWB = 2 * K + 16;                    // padded LDS row of a weight chunk (bytes): 784 = 196 dwords = 4 mod 64 banks
constexpr int CHUNK_PIECES = 28;                    // 32 rows x 784 B = 24.5 KB, padded so that every wave issues 7 pieces
constexpr int CHUNKB = CHUNK_PIECES * 1024;
constexpr int NPW = CHUNK_PIECES / 4;
constexpr int NBUF = 4;                             // LDS ring: unit u multiplies buffer u % NBUF; chunk u + NBUF - 1 is requested at its top
constexpr int DREQ = NBUF - 1;                      // (a fifth buffer, or fragments read two or three k-steps ahead, changed nothing:
constexpr int WPF = 1;                              //  profiles/r03_rc64_ablations.log) k-steps a weight fragment is read ahead of its MFMAs
constexpr int NPWE = CHUNK_PIECES / 4;                  // vector-memory operations a wave issues per unit: DMA pieces ...
constexpr int ST_Q3 = 8 + 8;                            // ... c_prev loads + c' stores under the o unit (h' stores under i: 0 or 4)
#ifndef RC64_PIECE_EVERY
#define RC64_PIECE_EVERY 6
#endif
// A wave's DMA pieces are dealt out over the unit, one per PIECE_EVERY MFMAs: vector-memory instructions issued back to back
// hold the wave for 300-400 cycles apiece once the CU's memory queue is full (profiles/r03_rc64_gap_stamps.log; the same
// finding as lstm_mfma64.hip's DMA_EVERY).
constexpr int PIECE_EVERY = RC64_PIECE_EVERY;
static_assert((NPWE - 1) * PIECE_EVERY < 2 * KSTEPS, "the request must fit into the unit");
// at the end of unit u chunk u + 1 (requested during unit u - 2) must have landed: what a wave has certainly issued after that
// chunk's last piece may stay in flight -- the pieces of units u - 1 and u and their state words, the h' stores left out (a
// smaller count waits for more, never for less)
constexpr int WAIT_END_Q03 = (DREQ - 1) * NPWE + ST_Q3;     // i unit: the previous o unit's words ; o unit: its own
constexpr int WAIT_END_Q12 = (DREQ - 1) * NPWE;             // g and f units: no state words in them or before them
static_assert(DREQ == 3, "the wait counts above are written for a four-deep ring");
constexpr int AX = 0, AH = 64, ACP = 192;           // accumulator-register regions (inline assembly only)
constexpr float L2E = 1.4426950408889634f;

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int V> using IC = std::integral_constant<int, V>;

// Every hand-written vector-memory instruction below takes its address from a per-lane 64-bit VGPR pointer, never from an SGPR
// pair: when the compiler spills SGPRs it restores them with v_readlane, and a VALU-written SGPR read by a VMEM instruction
// within five wait states is stale -- the hazard recognizer inserts those wait states for its own VMEM instructions, not for
// the ones inside inline assembly (first version of this kernel: a garbage address, a GPU memory fault).
__device__ __forceinline__ void dma_piece(const uint8_t* src_lane, uint32_t lds_off)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_off), "v"(src_lane) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int LO> __device__ __forceinline__ void mfma_a(f32x16_t& acc, const bf16x8_t& w)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma_a0(f32x16_t& acc, const bf16x8_t& w)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(acc) : "v"(w), "n"(LO), "n"(LO + 3));
}
// 16 bytes per lane at (per-lane pointer + IMM) -> a[LO : LO + 3]; IMM in [0, 4096)
template <int LO, int IMM> __device__ __forceinline__ void ld_a4(const uint8_t* p)
{
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3" :: "v"(p), "n"(LO), "n"(LO + 3), "n"(IMM) : "memory");
}
template <int IMM> __device__ __forceinline__ void st_v4(uint8_t* p, const u32x4_t& v)
{
    // the s_nop: a store of more than 8 bytes reads its data registers over several cycles, and a VALU instruction that
    // overwrites them in the very next slot is seen by the last lanes (gfx9 "VMEM store data" hazard, one wait state) -- the
    // compiler pads its own stores, not this one (second version of this kernel: h' wrong in lanes 12-15 / 28-31 of each half)
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 0" :: "v"(p), "v"(v), "n"(IMM) : "memory");
}
// fragment F (1 KB each) of a lane's stream: pointer of its group of four + immediate
template <int LO, int F> __device__ __forceinline__ void ld_frag(const uint8_t* lane_base) { ld_a4<LO, (F & 3) * 1024>(lane_base + (F >> 2) * 4096); }
template <int A> __device__ __forceinline__ float rd_a()
{
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(A));
    return v;
}
template <int A> __device__ __forceinline__ void zero_a() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(A)); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t p = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// B fragment (row tile t, k-step ks) of the concatenated input [feats | h]
constexpr int slab_reg(int t, int ks) { return ks < XSTEPS ? AX + 4 * (XSTEPS * t + ks) : AH + 4 * (HSTEPS * t + (ks - XSTEPS)); }



struct CellPtrs { const uint8_t* h_in; const uint8_t* c_in; uint8_t* h_out; uint8_t* c_out; };

__global__ void __launch_bounds__(256, 1)
policy_rc64_kernel(const uint8_t* __restrict__ feats /*fragment layout [B/64][2][8][64][16 B]*/, const float* __restrict__ keep /*[B]*/,
                   const uint8_t* __restrict__ wimg /*[2][8][4] chunks of CHUNKB*/, const float* __restrict__ bias /*[2][4H] i,f,g,o*/,
                   CellPtrs actor, CellPtrs critic)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_w[NBUF * CHUNKB];
    __shared__ __attribute__((aligned(16))) float s_b[2 * 4 * H];         // pre-scaled for the exponent forms

    asm volatile("" ::: "a0", "a223");                  // a[0:223] belong to the inline assembly below

    const int tid = threadIdx.x, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int64_t wb = int64_t(blockIdx.x) * 4 + uwave;                   // this wave's block of 64 rows
    const uint32_t lane16 = uint32_t(lane * 16);
    const uint32_t lds_w = uint32_t(reinterpret_cast<uintptr_t>(&s_w[0]));

    // episode-start masks of this lane's two rows (tile 0: row r, tile 1: row 32 + r)
    const float k0 = keep ? keep[wb * 64 + r] : 1.0f, k1 = keep ? keep[wb * 64 + 32 + r] : 1.0f;

    // ---- weight stream: chunk c (global unit index) -> LDS buffer c % 4; this wave's pieces uwave, uwave + 4, ...
    const uint8_t* wimg_lane = wimg + uwave * 1024 + lane16;
    auto req_piece = [&](int slot, int chunk) {
        dma_piece(wimg_lane + int64_t(chunk) * CHUNKB + slot * 4096, lds_w + uint32_t((chunk % NBUF) * CHUNKB + (uwave + 4 * slot) * 1024));
    };
#pragma unroll
    for (int c = 0; c < DREQ; ++c)
#pragma unroll
        for (int s = 0; s < NPW; ++s) req_piece(s, c);

    for (int i = tid; i < 2 * 4 * H; i += 256) {
        const int gate = (i % (4 * H)) / H;
        s_b[i] = bias[i] * (gate == 2 ? -2.0f * L2E : -L2E);
    }

    // ---- feature fragments (shared by both cells) and the actor's h fragments and first c_prev slice
    {
        const uint8_t* fb = feats + wb * (2 * XSTEPS * 1024) + lane16;
        sfor<0, 2 * XSTEPS>([&](auto F) { constexpr int f = decltype(F)::value; ld_frag<AX + 4 * f, f>(fb); });
    }
    auto load_h = [&](const uint8_t* h_in) {
        const uint8_t* hb = h_in + wb * (2 * HSTEPS * 1024) + lane16;
        sfor<0, 2 * HSTEPS>([&](auto F) { constexpr int f = decltype(F)::value; ld_frag<AH + 4 * f, f>(hb); });
    };
    // this lane's 16 bytes of (slice sl, tile 0, group 0) of a cell state; group (t, q) sits (4 t + q) KB further on
    auto c_slice = [&](const uint8_t* c, int sl) { return c + (wb * NSLICE + sl) * (2 * 4 * 1024) + lane16; };
    load_h(actor.h_in);
    {
        const uint8_t* cb = c_slice(actor.c_in, 0);
        sfor<0, 8>([&](auto G) { constexpr int g = decltype(G)::value; ld_frag<ACP + 4 * g, g>(cb); });
    }
    // rows whose episode just started take h = 0 (rare; after the loads have landed)
    auto mask_h = [&]() {
        if (k0 == 0.0f) sfor<0, 4 * HSTEPS>([&](auto A) { zero_a<AH + decltype(A)::value>(); });
        if (k1 == 0.0f) sfor<0, 4 * HSTEPS>([&](auto A) { zero_a<AH + 4 * HSTEPS + decltype(A)::value>(); });
    };

    wait_vm<2 * HSTEPS + 8>();                          // chunks 0..2 and the feature fragments have landed (younger: h, c_prev)
    __syncthreads();                                    // ... for every wave; the biases are in LDS

    auto wfrag = [&](int buf, int ks) {                 // A operand: lane (n = r, hf) reads 16 bytes of chunk row r, k-step ks
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(&s_w[buf * CHUNKB + r * ROWB + ks * 32 + hf * 16]));
    };
    int ubuf = 0;                                       // LDS buffer of the unit about to run (= its global index mod NBUF)

    f32x16_t acc[2][2];                                 // [set][tile]: units alternate sets; the other set is being consumed
    float V[2][16];                                     // the slice's running values: I -> i g -> c' -> E = 2^(-2 c' log2 e)
    uint32_t hp[2][8];                                  // packed h' of the pending slice

    // state of the slice whose o epilogue is pending
    int p_sl = 0, p_cell = 0;
    uint8_t* p_hlane = actor.h_out;                     // this lane's 16 bytes of fragment (tile 0, k-step 2 p_sl) of the pending slice's h'

    // one unit: 48 MFMAs of gate position Q (LDS buffer Q) into acc[Q & 1]; after MFMA m: this wave's DMA piece m of chunk
    // `next_chunk` (m < NPW), then micro-stages [m U / 48, (m + 1) U / 48) of `micro`.  CELLSTART (the i unit of a cell's slice 0):
    // the h fragments are still arriving -- wait and mask before k-step XSTEPS.
    f32x4_t bq[4];                                      // the unit's epilogue biases (same units for both row tiles), read at its top
    auto unit = [&](auto q_c, auto nmicro_c, auto&& micro, int next_chunk, bool cellstart, int bias_off) {
        constexpr int Q = decltype(q_c)::value, U = decltype(nmicro_c)::value, M = 2 * KSTEPS;
        bf16x8_t w[WPF + 1];
        const int buf = ubuf;
        ubuf = ubuf + 1 == NBUF ? 0 : ubuf + 1;
#pragma unroll
        for (int j = 0; j < WPF; ++j) w[j] = wfrag(buf, j);
        if constexpr (U > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4_t*>(&s_b[bias_off + 8 * q + 4 * hf]);
        }
        FENCE();
        sfor<0, KSTEPS>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            if constexpr (Q == 0 && ks == XSTEPS) {
                if (cellstart) { wait_vm<0>(); mask_h(); }
                FENCE();
            }
            sfor<0, 2>([&](auto T) {
                constexpr int t = decltype(T)::value, m = 2 * ks + t;
                if constexpr (t == 0 && ks + WPF < KSTEPS) w[(ks + WPF) % (WPF + 1)] = wfrag(buf, ks + WPF);
                if constexpr (ks == 0) mfma_a0<slab_reg(t, ks)>(acc[Q & 1][t], w[ks % (WPF + 1)]);
                else mfma_a<slab_reg(t, ks)>(acc[Q & 1][t], w[ks % (WPF + 1)]);
                if constexpr (m % PIECE_EVERY == 0 && m / PIECE_EVERY < NPWE) req_piece(m / PIECE_EVERY, next_chunk);
                if constexpr (m >= 1) sfor<((m - 1) * U) / (M - 1), (m * U) / (M - 1)>([&](auto UU) { micro(UU); });
                FENCE();
            });
        });
        asm volatile("s_nop 15\n\ts_nop 3");            // MFMA result -> VALU read distance (the compiler cannot see into the asm)
    };

    struct GS { float x[4], y[4], z[4]; };
    GS gs[8];
    // micro index u -> (group gq = (tile, q), stage st): the two groups of a pair alternate, dependent operations sit apart
    #define RC_DECODE(NS) constexpr int u = decltype(UU)::value, pr = u / (2 * (NS)), st = (u % (2 * (NS))) / 2, gq = 2 * pr + (u & 1); \
                          constexpr int t = gq >> 2, q = gq & 3; GS& s = gs[gq]; (void)t; (void)q; (void)s;

    for (int g = 0; g < 2 * NSLICE; ++g) {
        const int cell = g >> 3, sl = g & 7;
        const CellPtrs& cp = cell ? critic : actor;
        const int u0 = 4 * g;                            // global index of this slice's first unit
        const int sb = cell * 4 * H + 32 * sl;           // + gate * H + 8 q + 4 hf
        const uint32_t kv[2] = { __builtin_bit_cast(uint32_t, k0), __builtin_bit_cast(uint32_t, k1) };

        // ---- unit 0: gate i -> set 0 ; under it the o epilogue of the PENDING slice: h' = sigmoid(o) tanh(c') -> pack -> store
        constexpr int NS_O = 8;
        auto p_o = [&](auto UU) {
            RC_DECODE(NS_O)
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = __builtin_fmaf(acc[1][t][4 * q + i], -L2E, bq[q][i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = ex2(s.x[i]); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = 1.0f + V[t][4 * q + i]; });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.z[i], s.x[i], s.z[i]); });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = rcp(s.z[i]); });
            } else if constexpr (st == 5) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = 1.0f - V[t][4 * q + i]; });
            } else if constexpr (st == 6) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = s.y[i] * s.z[i]; });
            } else {
                hp[t][2 * q] = pack2(s.y[0], s.y[1]); hp[t][2 * q + 1] = pack2(s.y[2], s.y[3]);
                if constexpr (q & 1) {                  // fragment hq = q >> 1 of tile t is complete: k-step 2 p_sl + hq
                    constexpr int hq = q >> 1;
                    const u32x4_t v = { hp[t][4 * hq], hp[t][4 * hq + 1], hp[t][4 * hq + 2], hp[t][4 * hq + 3] };
                    st_v4<hq * 1024>(p_hlane + t * (HSTEPS * 1024), v);
                }
            }
        };
        if (g == 0) unit(IC<0>{}, IC<0>{}, [&](auto) {}, (u0 + DREQ) % NUNITS, true, 0);
        else unit(IC<0>{}, IC<8 * NS_O>{}, p_o, (u0 + DREQ) % NUNITS, sl == 0, p_cell * 4 * H + 3 * H + 32 * p_sl);
        wait_vm<WAIT_END_Q03>();                        // chunk u0 + 1 has landed
        __syncthreads();

        // ---- unit 1: gate g -> set 1 ; under it I = 2^(-i log2 e) from set 0
        constexpr int NS_I = 2;
        auto p_i = [&](auto UU) {
            RC_DECODE(NS_I)
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = __builtin_fmaf(acc[0][t][4 * q + i], -L2E, bq[q][i]); });
            } else { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; V[t][4 * q + i] = ex2(s.x[i]); }); }
        };
        unit(IC<1>{}, IC<8 * NS_I>{}, p_i, (u0 + 1 + DREQ) % NUNITS, false, sb);
        wait_vm<WAIT_END_Q12>();
        __syncthreads();

        // ---- unit 2: gate f -> set 0 ; under it i g = sigmoid(i) tanh(g) = (1 - G) / ((1 + G)(1 + I)), G = 2^(-2 g log2 e), from set 1
        constexpr int NS_G = 8;
        auto p_g = [&](auto UU) {
            RC_DECODE(NS_G)
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = __builtin_fmaf(acc[1][t][4 * q + i], -2.0f * L2E, bq[q][i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = __builtin_amdgcn_fmed3f(s.y[i], -40.0f, 40.0f); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = ex2(s.y[i]); });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = 1.0f + s.y[i]; });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.z[i], V[t][4 * q + i], s.z[i]); });
            } else if constexpr (st == 5) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = rcp(s.z[i]); });
            } else if constexpr (st == 6) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = 1.0f - s.y[i]; });
            } else { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; V[t][4 * q + i] = s.y[i] * s.z[i]; }); }
        };
        unit(IC<2>{}, IC<8 * NS_G>{}, p_g, (u0 + 2 + DREQ) % NUNITS, false, sb + 2 * H);
        wait_vm<WAIT_END_Q12>();
        __syncthreads();

        // ---- unit 3: gate o -> set 1 ; under it c' = sigmoid(f) keep c + i g from set 0 -> store, then E = 2^(-2 c' log2 e);
        // each group's c_prev registers are re-requested for the NEXT slice as soon as they have been read
        const int gn = g + 1 < 2 * NSLICE ? g + 1 : g;
        const uint8_t* cnext = c_slice((gn >> 3) ? critic.c_in : actor.c_in, gn & 7);
        uint8_t* cout = cp.c_out + (wb * NSLICE + sl) * (2 * 4 * 1024) + lane16;
        constexpr int NS_F = 11;
        auto p_f = [&](auto UU) {
            RC_DECODE(NS_F)
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = __builtin_fmaf(acc[0][t][4 * q + i], -L2E, bq[q][i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = ex2(s.x[i]); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = 1.0f + s.x[i]; });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = rcp(s.x[i]); });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = rd_a<ACP + 16 * t + 4 * q + i>(); });
            } else if constexpr (st == 5) {
                sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] *= __builtin_bit_cast(float, kv[t]); });
                ld_a4<ACP + 16 * t + 4 * q, q * 1024>(cnext + t * 4096);
            } else if constexpr (st == 6) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; V[t][4 * q + i] = __builtin_fmaf(s.y[i], s.x[i], V[t][4 * q + i]); });
            } else if constexpr (st == 7) {
                const u32x4_t v = { __builtin_bit_cast(uint32_t, V[t][4 * q]), __builtin_bit_cast(uint32_t, V[t][4 * q + 1]),
                                    __builtin_bit_cast(uint32_t, V[t][4 * q + 2]), __builtin_bit_cast(uint32_t, V[t][4 * q + 3]) };
                st_v4<q * 1024>(cout + t * 4096, v);
            } else if constexpr (st == 8) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = V[t][4 * q + i] * (-2.0f * L2E); });
            } else if constexpr (st == 9) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_amdgcn_fmed3f(s.z[i], -40.0f, 40.0f); });
            } else { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; V[t][4 * q + i] = ex2(s.z[i]); }); }
        };
        // past the last chunk the request wraps to the image's first chunks (harmless: nobody multiplies them)
        unit(IC<3>{}, IC<8 * NS_F>{}, p_f, (u0 + 3 + DREQ) % NUNITS, false, sb + H);
        p_sl = sl; p_cell = cell; p_hlane = cp.h_out + (wb * (2 * HSTEPS) + 2 * sl) * 1024 + lane16;
        if (g == NSLICE - 1) {                          // the actor is done with its h fragments: the critic's may come in
            load_h(critic.h_in);
            wait_vm<(WAIT_END_Q03 + 2 * HSTEPS > 63 ? 63 : WAIT_END_Q03 + 2 * HSTEPS)>();
        } else {
            wait_vm<WAIT_END_Q03>();
        }
        __syncthreads();
    }
    #undef RC_DECODE
    // ---- drain: the last slice's o epilogue has nothing to hide under
    {
        constexpr int NS_O = 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4_t*>(&s_b[p_cell * 4 * H + 3 * H + 32 * p_sl + 8 * q + 4 * hf]);
        sfor<0, 8 * NS_O>([&](auto UU) {
            constexpr int u = decltype(UU)::value, pr = u / (2 * NS_O), st = (u % (2 * NS_O)) / 2, gq = 2 * pr + (u & 1);
            constexpr int t = gq >> 2, q = gq & 3;
            GS& s = gs[gq];
            if constexpr (st == 0) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = __builtin_fmaf(acc[1][t][4 * q + i], -L2E, bq[q][i]); });
            } else if constexpr (st == 1) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.x[i] = ex2(s.x[i]); });
            } else if constexpr (st == 2) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = 1.0f + V[t][4 * q + i]; });
            } else if constexpr (st == 3) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = __builtin_fmaf(s.z[i], s.x[i], s.z[i]); });
            } else if constexpr (st == 4) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.z[i] = rcp(s.z[i]); });
            } else if constexpr (st == 5) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = 1.0f - V[t][4 * q + i]; });
            } else if constexpr (st == 6) { sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.y[i] = s.y[i] * s.z[i]; });
            } else {
                hp[t][2 * q] = pack2(s.y[0], s.y[1]); hp[t][2 * q + 1] = pack2(s.y[2], s.y[3]);
                if constexpr (q & 1) {
                    constexpr int hq = q >> 1;
                    const u32x4_t v = { hp[t][4 * hq], hp[t][4 * hq + 1], hp[t][4 * hq + 2], hp[t][4 * hq + 3] };
                    st_v4<hq * 1024>(p_hlane + t * (HSTEPS * 1024), v);
                }
            }
        });
    }
    wait_vm<0>();                                       // the wrapped chunk requests must not outlive the workgroup's LDS
}

}  // namespace


extern "C" int fdyn_policy_recurrent_image_bytes(void) { return 2 * NSLICE * 4 * CHUNKB; }

extern "C" int fdyn_policy_recurrent(const void* feats_frag, const float* keep, const void* weight_image, const float* bias,
                                     const void* h_pi_in, const float* c_pi_in, void* h_pi_out, float* c_pi_out,
                                     const void* h_vf_in, const float* c_vf_in, void* h_vf_out, float* c_vf_out,
                                     int64_t B, void* stream)
{
    if (!feats_frag || !weight_image || !bias || !h_pi_in || !c_pi_in || !h_pi_out || !c_pi_out || !h_vf_in || !c_vf_in || !h_vf_out ||
        !c_vf_out) return FDYN_ERR_NULL;
    if (B <= 0 || B % 256) return FDYN_ERR_BAD_SIZE;
    CellPtrs a = { (const uint8_t*)h_pi_in, (const uint8_t*)c_pi_in, (uint8_t*)h_pi_out, (uint8_t*)c_pi_out };
    CellPtrs c = { (const uint8_t*)h_vf_in, (const uint8_t*)c_vf_in, (uint8_t*)h_vf_out, (uint8_t*)c_vf_out };
    hipLaunchKernelGGL(policy_rc64_kernel, dim3(unsigned(B / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)feats_frag, keep,
                       (const uint8_t*)weight_image, bias, a, c);
    return int(hipGetLastError());
}
