// lstm_mfma64.hip -- the recurrent LSTM cell at ONE wave per SIMD: 64 batch rows per wave, the whole 512-register file.
//
// Same cell as lstm_mfma.hip (nn.LSTM's step: learned_controllers/networks/lstm_policy.py:49-61 and sb3_contrib's actor /
// critic LSTMs), same operand roles (activations = MFMA A operand in registers, weights = B operand streamed through LDS),
// different shape.  In lstm_mfma.hip a wave owns 32 rows, so every B fragment read from LDS feeds ONE MFMA, and with 256
// registers per wave the compiler had no room to request a fragment ahead of its use: each MFMA paid the LDS latency
// (matrix pipe 26 % busy whatever else was tried).  Here:
//   * workgroup = 4 waves = 256 batch rows, one workgroup per CU, one wave per SIMD, 512 registers per lane;
//   * a wave keeps TWO 32-row slabs and every B fragment feeds two MFMAs: half the LDS reads and half the L2 -> LDS weight
//     stream per flop (201 MB instead of 403 MB per 65 536-row cell), four independent accumulators per k-step;
//   * a chunk is a whole unit (gate pair x 32 hidden units x all K = 384 columns, 49 KB padded): one barrier per 96 MFMAs
//     instead of one per 24; two LDS buffers, chunk q + 1 requested by LDS-DMA at the top of chunk q's block;
//   * the register file is split BY HAND: the slabs (192 registers), the cell state on its way in (32) live in accumulator
//     registers a[18:254] that only inline assembly touches (MFMA A operands and load destinations may be AGPRs on gfx950);
//     the compiler owns the 256 architectural registers (two accumulator sets 128, sigmoid(i) tanh(g) 32, B fragments in
//     flight 24, epilogue temporaries).  Left to the compiler (MFMA builtins) the accumulators went to AGPRs and every
//     epilogue read became a copy, 53 registers spilled to scratch;
//   * the point-wise epilogue is software-pipelined under the next unit's MFMAs (two accumulator sets, as in
//     lstm_mfma.hip), but placed BY HAND: with one wave per SIMD nothing else fills the issue slots an MFMA leaves free
//     (24 of its 32 cycles), and sched_group_barrier patterns were not honoured (transcendentals are not in its VALU
//     class: they and everything that depends on them sank below the last MFMA).  Every MFMA is followed in the source by
//     its share of the epilogue -- one STAGE of four elements (e.g. the four exp2 of sigmoid(f)) -- and a scheduling fence.
// Full 256-row blocks with c' wanted and no fp32 copy of h' only (the rollout's configuration); everything else stays
// with lstm_mfma.hip (the C entry point fdyn_lstm_cell_mfma chooses).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/fdyn.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM64 = 256;          // batch rows per workgroup
constexpr int NSL = 32;            // hidden units per slice
constexpr int PAD = 8;             // bf16 elements (16 B) of LDS row padding

// accumulator-register map (inline assembly only)
// a[0 : AG_BASE - 1] are left to the compiler: when it runs out of architectural registers it parks values in the lowest
// accumulator registers, and nothing can tell it that others are taken -- __graft_entry__.build() scans the ISA of this file
// and refuses a build in which compiler-generated code touches a[AG_BASE] or above.
constexpr int AG_BASE = 18;
constexpr int AG_SLAB = AG_BASE;          // a[AG_SLAB + 96 t + 4 ks .. + 3]: slab fragment (row tile t, k-step ks)
constexpr int AG_CP = AG_BASE + 192;      // a[AG_CP + 16 t + e]: c_prev of the slice whose (f, o) epilogue is pending
constexpr int AG_SOFF = AG_BASE + 224;    // a[AG_SOFF + j]: lane byte offset of DMA piece j inside a weight chunk

__device__ __attribute__((aligned(16))) uint32_t fdyn_zero_row[128];      // 512 zero bytes: the h row of an episode that just started

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// 64 lanes x 16 bytes from global memory (scalar base + 32-bit lane byte offset) to 1 KB of LDS at byte offset `lds_off`
// (wave-uniform).  Inline assembly for the reason given in lstm_mfma.hip: the compiler must not count it.
// the lane offset is kept in accumulator register a[A] (13 offsets are 13 architectural registers the epilogue needs)
template <int A> __device__ __forceinline__ void dma16a(const void* base, uint32_t lds_off)
{
    uint32_t t;
    asm volatile("v_accvgpr_read_b32 %0, a[%3]\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                 : "=&v"(t) : "s"(lds_off), "s"(base), "n"(A) : "memory");
}
template <int A> __device__ __forceinline__ void wr_a(uint32_t v) { asm volatile("v_accvgpr_write_b32 a[%1], %0" :: "v"(v), "n"(A)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N > 60 ? 60 : (N < 0 ? 0 : N)) : "memory"); }

// acc += A(a[LO:LO+3]) x b ; the `zero` form starts a chain (C operand = inline constant 0)
template <int LO> __device__ __forceinline__ void mfma_acc(f32x16_t& acc, const bf16x8_t& b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%2:%3], %1, %0" : "+v"(acc) : "v"(b), "n"(LO), "n"(LO + 3));
}
template <int LO> __device__ __forceinline__ void mfma_zero(f32x16_t& acc, const bf16x8_t& b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%2:%3], %1, 0" : "=&v"(acc) : "v"(b), "n"(LO), "n"(LO + 3));
}
template <int LO, int IMM> __device__ __forceinline__ void ld_a4_s(const void* base, uint32_t voff)
{
    asm volatile("global_load_dwordx4 a[%2:%3], %0, %1 offset:%4" :: "v"(voff), "s"(base), "n"(LO), "n"(LO + 3), "n"(IMM) : "memory");
}
template <int LO, int IMM> __device__ __forceinline__ void ld_a4_v(const void* vaddr)
{
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3" :: "v"(vaddr), "n"(LO), "n"(LO + 3), "n"(IMM) : "memory");
}
template <int A, int IMM> __device__ __forceinline__ void ld_a1_s(const void* base, uint32_t voff)
{
    asm volatile("global_load_dword a[%2], %0, %1 offset:%3" :: "v"(voff), "s"(base), "n"(A), "n"(IMM) : "memory");
}
template <int A> __device__ __forceinline__ float rd_a()
{
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(A));
    return v;
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
constexpr float L2E = 1.4426950408889634f;

// One cell's operands.  A launch takes TWO sets (blockIdx.y picks one): the policy's actor and critic cells read the same x and are
// independent, so one launch of 2 x B / 256 workgroups runs them back to back on every CU -- no second launch, and a CU that is
// done with its actor block starts its critic block while slower CUs finish.
struct CellArgs {
    const uint16_t* x; const uint16_t* h_prev; const float* c_prev; const float* keep;      // [B][KX] bf16, [B][KH] bf16, [B][H], [B] or null
    const uint16_t* W; const float* bias; uint16_t* h_out; float* c_out;                  // [4H][KX+KH] bf16, [4H], [B][H] bf16, [B][H]
};

template <int KX, int KH, int H>
__global__ void __launch_bounds__(256, 1)
lstm_cell_mfma64_kernel(CellArgs a0, CellArgs a1)
{
    const bool second = blockIdx.y != 0;                  // scalar selects
    const uint16_t* __restrict__ x = second ? a1.x : a0.x;
    const uint16_t* __restrict__ h_prev = second ? a1.h_prev : a0.h_prev;
    const float* __restrict__ c_prev = second ? a1.c_prev : a0.c_prev;
    const float* __restrict__ keep = second ? a1.keep : a0.keep;
    const uint16_t* __restrict__ W = second ? a1.W : a0.W;
    const float* __restrict__ bias = second ? a1.bias : a0.bias;
    uint16_t* __restrict__ h_out = second ? a1.h_out : a0.h_out;
    float* __restrict__ c_out = second ? a1.c_out : a0.c_out;
    constexpr int K = KX + KH;
    constexpr int KSTEPS = K / 16;
    constexpr int XSTEPS = KX / 16;
    constexpr int ROW = K + PAD;                          // padded LDS row (bf16 elements): 784 B, 196 dwords = 4 mod 64 banks
    constexpr int SPR = ROW / 8;                          // 16-byte slots per padded row (48 data + 1 pad)
    constexpr int NGROUPS = SPR;                          // 64 rows x SPR slots = SPR groups of 64 slots
    constexpr int NG = (NGROUPS + 3) / 4;                 // DMA instructions per chunk per wave, the same for every wave
    constexpr int BUF = 64 * ROW;                         // elements per chunk buffer
    constexpr int NSLICES = H / NSL;
    constexpr int PD = 1;                                 // k-steps a B fragment is requested ahead of its MFMAs
    constexpr int DMA_EVERY = 3;                          // one DMA instruction of the next chunk's request per this many MFMAs
    // VMEM bookkeeping of a block (all compile-time).  In gap g = 12 sg + m the order is MFMA, DMA piece (if any), filler.
    // P0 stores 4 + 4 words per super-group (m = 5, 10), P1 loads 4 words of c_prev per super-group (m = 0, 6, 9, 10).
    constexpr int LAST_DMA_GAP = (NG - 1) * DMA_EVERY;
    static_assert(LAST_DMA_GAP < 96, "the request must fit into the block");
    constexpr auto p0_ops_from = [](int g) { int n = 0; for (int q = g; q < 96; ++q) n += (q % 12 == 5 || q % 12 == 10) ? 4 : 0; return n; };
    constexpr auto p0_stores_before_sg = [](int sg) { return 8 * sg; };
    constexpr auto p1_ops_from = [](int g) { int n = 0; for (int q = g; q < 96; ++q) n += (q % 12 == 0 || q % 12 == 6 || q % 12 == 9 || q % 12 == 10) ? 1 : 0; return n; };
    constexpr int LAST_DMA_LATEST = LAST_DMA_GAP;
    constexpr int P0_AFTER_DMA = p0_ops_from(LAST_DMA_LATEST);   // stores certainly younger than the block's last DMA piece
    constexpr int P1_AFTER_DMA = p1_ops_from(LAST_DMA_LATEST);   // c_prev loads certainly younger than it
    __shared__ __attribute__((aligned(16))) uint16_t s_w[2 * BUF];
    __shared__ __attribute__((aligned(16))) float s_keep[BM64];
    __shared__ __attribute__((aligned(16))) float s_bias[4 * H];          // pre-scaled for the exponent FMAs

    asm volatile("" ::: "a18", "a254");                    // the accumulator registers a[0:223] belong to the inline assembly below

    const int tid = threadIdx.x, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int64_t urow0 = int64_t(blockIdx.x) * BM64 + uwave * 64;        // first row of this wave (wave-uniform)

    // ---- weight stream.  Slot v of a chunk = padded row v / SPR, 16-byte column v % SPR (the pad column re-reads the last data
    // column); the 64 lanes of one DMA instruction fill slots 64 g .. 64 g + 63; wave w issues groups (w + 4 j) mod NGROUPS.
    sfor<0, NG>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int g = (uwave + 4 * j) % NGROUPS;
        const int v = g * 64 + lane;
        const int row = v / SPR, cs = v % SPR;
        wr_a<AG_SOFF + j>(uint32_t((((row >> 5) * 2 * H + (row & 31)) * K + (cs < SPR - 1 ? cs : SPR - 2) * 8) * 2));
    });
    const int sl_start = int((blockIdx.x + (blockIdx.x >> 3)) % unsigned(NSLICES));      // rotated slice order (see lstm_mfma.hip)
    auto slice_of = [&](int i) { return (i + sl_start) % NSLICES; };
    const uint32_t lds_w = uint32_t(reinterpret_cast<uintptr_t>(&s_w[0]));
    auto request = [&](int sl, int pass, int buf) {       // chunk (slice sl, pass) -> LDS buffer buf
        const uint16_t* org = W + (int64_t(pass) * H + sl * NSL) * K;
        sfor<0, NG>([&](auto J) {
            constexpr int j = decltype(J)::value;
            dma16a<AG_SOFF + j>(org, lds_w + uint32_t(buf * BUF * 2 + ((uwave + 4 * j) % NGROUPS) * 1024));
        });
    };
    // ---- prologue.  Every global load of this kernel is inline assembly with hand-counted waits: one load the compiler knows
    // about would bring its own s_waitcnt, computed without the DMA and slab loads in flight around it.
    //   order in the queue: keep (1), bias (1), chunk 0 (NG), chunk 1 (NG), x part of both slabs (2 XSTEPS), h part (2 HSTEPS)
    constexpr int HSTEPS = KSTEPS - XSTEPS;
    float kv;                                             // episode-start flag of row urow0 + lane (= row tid of the block)
    f32x4_t bv;
    {
        const float* kaddr = keep ? keep + urow0 + lane : reinterpret_cast<const float*>(fdyn_zero_row);
        asm volatile("global_load_dword %0, %1, off" : "=v"(kv) : "v"(kaddr) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv) : "v"(bias + 4 * tid) : "memory");
    }
    request(slice_of(0), 0, 0);
    request(slice_of(0), 1, 1);
    // piece J of a chunk (one DMA instruction of this wave), for the requests that are dealt out between the MFMAs of a block
    auto piece = [&](auto J, const uint16_t* org, int buf) {
        constexpr int j = decltype(J)::value;
        dma16a<AG_SOFF + j>(org, lds_w + uint32_t(buf * BUF * 2 + ((uwave + 4 * j) % NGROUPS) * 1024));
    };
    auto origin = [&](int sl, int pass) { return W + (int64_t(pass) * H + sl * NSL) * K; };
    {
        const uint16_t* xa = x + urow0 * KX;              // wave-uniform bases, one lane offset for both tiles
        const uint16_t* xb = x + (urow0 + 32) * KX;
        const uint32_t xoff = uint32_t((r * KX + 8 * hf) * 2);
        sfor<0, XSTEPS>([&](auto KS) {                    // fragment order (ks, tile): k-step ks may start once 2 ks + 2 have landed
            constexpr int ks = decltype(KS)::value;
            ld_a4_s<AG_SLAB + 4 * ks, 32 * ks>(xa, xoff);
            ld_a4_s<AG_SLAB + 96 + 4 * ks, 32 * ks>(xb, xoff);
        });
        // keep and bias have landed (and with them both chunks: they are older than the x loads)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(kv), "+v"(bv) : "n"(2 * XSTEPS) : "memory");
        if (!keep) kv = 1.0f;
        s_keep[tid] = kv;
        static_assert(4 * H == 4 * 256, "one float4 of bias per thread");
        const float sc = (4 * tid) / H == 2 ? 2.0f * L2E : -L2E;          // pre-scaled: gate g feeds tanh, the others sigmoid
        bv.x *= sc; bv.y *= sc; bv.z *= sc; bv.w *= sc;
        *reinterpret_cast<f32x4_t*>(&s_bias[4 * tid]) = bv;
        // the h part of a row whose episode just started is read from a row of zeros
        const uint64_t alive = __ballot(kv != 0.0f);      // bit l = row urow0 + l
        const bool ka = (alive >> r) & 1, kb = (alive >> (32 + r)) & 1;
        const char* zrow = reinterpret_cast<const char*>(fdyn_zero_row) + 16 * hf;
        const char* ha = ka ? reinterpret_cast<const char*>(h_prev + (urow0 + r) * KH + 8 * hf) : zrow;
        const char* hb = kb ? reinterpret_cast<const char*>(h_prev + (urow0 + 32 + r) * KH + 8 * hf) : zrow;
        sfor<0, HSTEPS>([&](auto KS) {
            constexpr int ks = XSTEPS + decltype(KS)::value;
            ld_a4_v<AG_SLAB + 4 * ks, 32 * (ks - XSTEPS)>(ha);
            ld_a4_v<AG_SLAB + 96 + 4 * ks, 32 * (ks - XSTEPS)>(hb);
        });
    }
    __syncthreads();                                      // chunks, mask and biases are visible to every wave

    const uint32_t uoff = uint32_t(hf * 4 * H);
    f32x16_t aA[2][2], aB[2][2];                          // [gate of the pair][row tile]: set A = (i, g), set B = (f, o)
    float ig[2][16];
    uint32_t pcol = 0;                                    // hidden unit (column) of the slice whose (f, o) epilogue is pending
    float pbo = 0.0f, pbf = 0.0f;

    // B fragments of k-step ks: lane (r, hf) reads 16 bytes of padded row r (first gate) / 32 + r (second gate)
    const uint16_t* wlane = s_w + r * ROW + hf * 8;
    auto bfrag = [&](int buf, int gate, int ks) {
        return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wlane + buf * BUF + gate * NSL * ROW + ks * 16));
    };

    // All MFMAs of one unit from LDS buffer BUFI into acc[2][2]: eight super-groups of three k-steps = 12 MFMAs.  MFMA m of
    // super-group sg (gap g = 12 sg + m) is followed by dma(piece g / DMA_EVERY) when g is a multiple of DMA_EVERY (the next
    // chunk's request, dealt out over the first 40 % of the block: thirteen DMA instructions back to back at the top of a block
    // cost each wave ~200 cycles apiece -- all four waves of the CU push 1 KB pieces through the one address path at once --
    // 22 of the kernel's 72 us), then by filler(sg, m), then by a scheduling fence.
    // SLABWAIT: the slab fragments are still arriving (first unit).
    auto unit = [&](auto bufc, f32x16_t (&acc)[2][2], auto slabwait_c, auto&& dma, auto&& filler) {
        constexpr int BUFI = decltype(bufc)::value;
        constexpr bool SLABWAIT = decltype(slabwait_c)::value;
        bf16x8_t p0[PD + 1], p1[PD + 1];
#pragma unroll
        for (int j = 0; j < PD; ++j) { p0[j] = bfrag(BUFI, 0, j); p1[j] = bfrag(BUFI, 1, j); }
        FENCE();
        sfor<0, KSTEPS / 3>([&](auto SG) {
            constexpr int sg = decltype(SG)::value;
            sfor<0, 3>([&](auto KK) {
                constexpr int kk = decltype(KK)::value;
                constexpr int ks = 3 * sg + kk;
                if constexpr (SLABWAIT) wait_vm<2 * KSTEPS - 2 * ks - 2>();
                const bf16x8_t b0 = p0[ks % (PD + 1)], b1 = p1[ks % (PD + 1)];
                sfor<0, 4>([&](auto MM) {
                    constexpr int mm = decltype(MM)::value;
                    constexpr int m = 4 * kk + mm, g = 12 * sg + m;
                    constexpr int alo = AG_SLAB + 96 * (mm & 1) + 4 * ks;
                    if constexpr (mm == 0 && ks + PD < KSTEPS) p0[(ks + PD) % (PD + 1)] = bfrag(BUFI, 0, ks + PD);
                    if constexpr (mm == 2 && ks + PD < KSTEPS) p1[(ks + PD) % (PD + 1)] = bfrag(BUFI, 1, ks + PD);
                    if constexpr (ks == 0) mfma_zero<alo>(acc[mm >> 1][mm & 1], mm < 2 ? b0 : b1);
                    else mfma_acc<alo>(acc[mm >> 1][mm & 1], mm < 2 ? b0 : b1);
                    if constexpr (g % DMA_EVERY == 0 && g / DMA_EVERY < NG) dma(std::integral_constant<int, g / DMA_EVERY>{});
                    filler(SG, std::integral_constant<int, m>{});
                    FENCE();
                });
            });
        });
        asm volatile("s_nop 15\n\ts_nop 3");              // MFMA result -> VALU read distance (the compiler cannot see into the asm)
    };

    // ---- the (f, o) epilogue of the pending slice, elements (tile t, rows 8 j .. 8 j + 3 (+ 4 hf)), one stage per MFMA gap.
    // C/D map of a 32x32 tile: column = lane & 31 (hidden unit), row = (e & 3) + 8 (e >> 2) + 4 hf.
    //   c' = sigmoid(f) keep c + ig ; h' = sigmoid(o) tanh(c')
    // (Tried: the 8 x 32 output tiles transposed through a per-wave LDS buffer and stored as ONE dwordx4 / dwordx2 instruction per
    // group instead of 4 + 4 word stores -- 16 instead of 64 store instructions per slice, the same lines: 70.5 us against 70.3.
    // What the stores cost is their lines, not their number.)
    struct FO { float kp[4], cp[4], a[4], b[4], c[4], t[4]; };
    auto fo_stage = [&](FO& s, auto waitcp_c, auto SG, auto M) {
        constexpr int sg = decltype(SG)::value, m = decltype(M)::value;
        constexpr bool WAITCP = decltype(waitcp_c)::value;
        constexpr int t = sg >> 2, j = sg & 3;
        float* cb = c_out + (urow0 + 32 * t + 8 * j) * H;                 // wave-uniform
        uint16_t* hb = h_out + (urow0 + 32 * t + 8 * j) * H;
        const uint32_t lo = uoff + pcol;
        if constexpr (m == 0) {
            {   // c_prev group sg has landed: younger than its last load (previous P1 block, gap (sg, 10)) are that block's later
                // loads and DMA pieces and this block's DMA pieces and stores so far
                constexpr int g10 = 12 * sg + 10;
                constexpr int p1_dma_after = NG - (g10 / DMA_EVERY + 1 < NG ? g10 / DMA_EVERY + 1 : NG);          // pieces in gaps > g10
                constexpr int p0_dma_before = (12 * sg) / DMA_EVERY + 1 < NG ? (12 * sg) / DMA_EVERY + 1 : NG;        // pieces in gaps <= 12 sg
                if constexpr (WAITCP) wait_vm<4 * (7 - sg) + p1_dma_after + p0_dma_before + p0_stores_before_sg(sg)>();
            }
            const float4 k4 = *reinterpret_cast<const float4*>(&s_keep[uwave * 64 + 32 * t + 8 * j + 4 * hf]);
            s.kp[0] = k4.x; s.kp[1] = k4.y; s.kp[2] = k4.z; s.kp[3] = k4.w;
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.cp[i] = rd_a<AG_CP + 16 * t + 4 * j + i>(); });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.a[i] = __builtin_fmaf(aB[0][t][4 * j + i], -L2E, pbf); });
        } else if constexpr (m == 1) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.a[i] = ex2(s.a[i]); });
        } else if constexpr (m == 2) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.a[i] = 1.0f + s.a[i]; });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.b[i] = __builtin_fmaf(aB[1][t][4 * j + i], -L2E, pbo); });
        } else if constexpr (m == 3) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.a[i] = rcp(s.a[i]); });          // sigmoid(f)
        } else if constexpr (m == 4) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.cp[i] *= s.kp[i]; });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.c[i] = __builtin_fmaf(s.a[i], s.cp[i], ig[t][4 * j + i]); });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = s.c[i] * (2.0f * L2E); });
        } else if constexpr (m == 5) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = ex2(s.t[i]); });
            sfor<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(cb) + (lo * 4u + uint32_t(i * H * 4))) = s.c[i];
            });
        } else if constexpr (m == 6) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = s.t[i] + 1.0f; });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.b[i] = ex2(s.b[i]); });
        } else if constexpr (m == 7) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = rcp(s.t[i]); });
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.b[i] = 1.0f + s.b[i]; });
        } else if constexpr (m == 8) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.b[i] = rcp(s.b[i]); });          // sigmoid(o)
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = __builtin_fmaf(-2.0f, s.t[i], 1.0f); });   // tanh(c')
        } else if constexpr (m == 9) {
            sfor<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; s.t[i] = s.t[i] * s.b[i]; });
        } else if constexpr (m == 10) {
            sfor<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                *reinterpret_cast<__bf16*>(reinterpret_cast<char*>(hb) + (lo * 2u + uint32_t(i * H * 2))) = static_cast<__bf16>(s.t[i]);
            });
        }
    };
    // ---- the (i, g) epilogue of the current slice: ig = sigmoid(i) tanh(g), in element pairs A = (0, 1), B = (2, 3) of the
    // group so that no gap carries more than one transcendental pair; the slice's c_prev loads ride along (a[AG_CP ..])
    struct IG { float xi[4], xg[4]; };
    auto ig_stage = [&](IG& s, float bi, float bg, uint32_t col, auto SG, auto M) {
        constexpr int sg = decltype(SG)::value, m = decltype(M)::value;
        constexpr int t = sg >> 2, j = sg & 3;
        auto pairop = [&](auto P, auto&& f) { constexpr int p = decltype(P)::value; f(std::integral_constant<int, 2 * p>{}); f(std::integral_constant<int, 2 * p + 1>{}); };
        using A_ = std::integral_constant<int, 0>; using B_ = std::integral_constant<int, 1>;
        auto xi_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xi[i] = __builtin_fmaf(aA[0][t][4 * j + i], -L2E, bi); };
        auto xg_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xg[i] = __builtin_fmaf(aA[1][t][4 * j + i], 2.0f * L2E, bg); };
        auto ei_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xi[i] = ex2(s.xi[i]); };
        auto eg_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xg[i] = ex2(s.xg[i]); };
        auto ai_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xi[i] = 1.0f + s.xi[i]; };
        auto ag_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xg[i] = s.xg[i] + 1.0f; };
        auto si_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xi[i] = rcp(s.xi[i]); };
        auto rg_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xg[i] = rcp(s.xg[i]); };
        auto tg_ = [&](auto I) { constexpr int i = decltype(I)::value; s.xg[i] = __builtin_fmaf(-2.0f, s.xg[i], 1.0f); };
        auto ig_ = [&](auto I) { constexpr int i = decltype(I)::value; ig[t][4 * j + i] = s.xi[i] * s.xg[i]; };
        const float* cpb = c_prev + (urow0 + 32 * t + 8 * j) * H;         // wave-uniform
        const uint32_t lo = (uoff + col) * 4u;
        if constexpr (m == 0) { pairop(A_{}, xi_); ld_a1_s<AG_CP + 16 * t + 4 * j + 0, 0>(cpb, lo); }
        else if constexpr (m == 1) { pairop(A_{}, ei_); pairop(A_{}, xg_); }
        else if constexpr (m == 2) { pairop(A_{}, eg_); pairop(A_{}, ai_); }
        else if constexpr (m == 3) { pairop(A_{}, si_); pairop(A_{}, ag_); }
        else if constexpr (m == 4) { pairop(A_{}, rg_); pairop(B_{}, xi_); }
        else if constexpr (m == 5) { pairop(B_{}, ei_); pairop(A_{}, tg_); }
        else if constexpr (m == 6) { pairop(B_{}, xg_); pairop(A_{}, ig_); ld_a1_s<AG_CP + 16 * t + 4 * j + 1, H * 4>(cpb, lo); }
        else if constexpr (m == 7) { pairop(B_{}, eg_); pairop(B_{}, ai_); }
        else if constexpr (m == 8) { pairop(B_{}, si_); pairop(B_{}, ag_); }
        else if constexpr (m == 9) { pairop(B_{}, rg_); ld_a1_s<AG_CP + 16 * t + 4 * j + 2, 2 * H * 4>(cpb, lo); }
        else if constexpr (m == 10) { pairop(B_{}, tg_); ld_a1_s<AG_CP + 16 * t + 4 * j + 3, 3 * H * 4>(cpb, lo); }
        else { pairop(B_{}, ig_); }
    };

    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    auto slice = [&](int si, auto pending_c) {
        constexpr bool PENDING = decltype(pending_c)::value;
        const int sl = slice_of(si);
        const uint32_t col = uint32_t(sl * NSL + r);
        // ================= P0: (i, g) of slice sl -> set A (LDS buffer 0), under it the (f, o) epilogue of the previous slice
        // and the request of this slice's (f, o) chunk -> buffer 1 (free since the last barrier)
        if constexpr (PENDING) {
            FO st;
            const uint16_t* org = origin(sl, 1);
            unit(C0{}, aA, std::false_type{}, [&](auto J) { piece(J, org, 1); }, [&](auto SG, auto M) { fo_stage(st, std::true_type{}, SG, M); });
            wait_vm<P0_AFTER_DMA>();                      // the (f, o) chunk is in LDS (the stores after its last piece may stay in flight)
        } else {
            unit(C0{}, aA, std::true_type{}, [&](auto) {}, [&](auto, auto) {});      // its (f, o) chunk was requested in the prologue
        }
        __syncthreads();
        // ================= P1: (f, o) of slice sl -> set B (LDS buffer 1), under it the (i, g) epilogue of this slice and the
        // request of the next slice's (i, g) chunk -> buffer 0 (past the end: a harmless duplicate)
        const uint16_t* norg = origin(slice_of(si + 1 < NSLICES ? si + 1 : si), 0);
        const float bi = s_bias[col], bg = s_bias[2 * H + col];
        IG st;
        unit(C1{}, aB, std::false_type{}, [&](auto J) { piece(J, norg, 0); }, [&](auto SG, auto M) { ig_stage(st, bi, bg, col, SG, M); });
        wait_vm<P1_AFTER_DMA>();                          // the next (i, g) chunk is in LDS (the c_prev loads after its last piece are younger)
        __syncthreads();
        pcol = col;
        pbo = s_bias[3 * H + col];
        pbf = s_bias[H + col];
    };

    slice(0, std::false_type{});
    for (int si = 1; si < NSLICES; ++si) slice(si, std::true_type{});
    // ---- drain: the last slice's (f, o) epilogue has nothing to hide under
    wait_vm<0>();
    {
        FO st;
        sfor<0, KSTEPS / 3>([&](auto SG) { sfor<0, 12>([&](auto M) { fo_stage(st, std::false_type{}, SG, M); }); });
    }
}

}  // namespace

// 1 = launched, 0 = shape not handled here (the caller falls back to lstm_mfma.hip), < 0 = HIP error
extern "C" int fdyn_lstm_cell_mfma64_try(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                                         const void* W, const float* bias, void* h_out, float* c_out, int64_t B, int H, void* stream)
{
    if (!(kx == 128 && kh == 256 && H == 256) || B <= 0 || B % BM64 || !c_out || !c_prev || !h_prev) return 0;
    const CellArgs a = { (const uint16_t*)x, (const uint16_t*)h_prev, c_prev, keep, (const uint16_t*)W, bias, (uint16_t*)h_out, c_out };
    hipLaunchKernelGGL((lstm_cell_mfma64_kernel<128, 256, 256>), dim3(unsigned(B / BM64)), dim3(256), 0, (hipStream_t)stream, a, a);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}

// two independent cells on the same x and keep (actor, critic) in ONE launch; same return convention
extern "C" int fdyn_lstm_cell_mfma64_pair_try(const void* x, int kx, const float* keep, int kh, int64_t B, int H,
                                              const void* h_prev0, const float* c_prev0, const void* W0, const float* bias0, void* h_out0, float* c_out0,
                                              const void* h_prev1, const float* c_prev1, const void* W1, const float* bias1, void* h_out1, float* c_out1,
                                              void* stream)
{
    if (!(kx == 128 && kh == 256 && H == 256) || B <= 0 || B % BM64) return 0;
    if (!c_out0 || !c_prev0 || !h_prev0 || !c_out1 || !c_prev1 || !h_prev1) return 0;
    const CellArgs a0 = { (const uint16_t*)x, (const uint16_t*)h_prev0, c_prev0, keep, (const uint16_t*)W0, bias0, (uint16_t*)h_out0, c_out0 };
    const CellArgs a1 = { (const uint16_t*)x, (const uint16_t*)h_prev1, c_prev1, keep, (const uint16_t*)W1, bias1, (uint16_t*)h_out1, c_out1 };
    hipLaunchKernelGGL((lstm_cell_mfma64_kernel<128, 256, 256>), dim3(unsigned(B / BM64), 2), dim3(256), 0, (hipStream_t)stream, a0, a1);
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
