// lstm_mfma.hip -- the recurrent cell of the rate-controller policy as ONE hand-written MFMA kernel for gfx950.
//
//   gates[b, n] = sum_k [x_b | keep_b * h_b][k] * W[n][k] + bias[n]      n in [0, 4H), PyTorch gate order i, f, g, o
//   c' = sigmoid(f) * keep_b * c + sigmoid(i) * tanh(g) ;  h' = sigmoid(o) * tanh(c')
//
// i.e. nn.LSTM's cell for one time step (the reference's policy: learned_controllers/networks/lstm_policy.py:49-61 and
// sb3_contrib's actor / critic LSTMs), batch 16 384 .. 65 536, H = 256, K = 128 + 256.  The un-fused path (hipBLASLt GEMM
// -> [B, 4H] gate tensor in HBM -> point-wise kernel) writes and re-reads 134 MB of gates per cell per step; here the
// gate tile never leaves the accumulators.
//
// Tiling for CDNA4 (wave64, v_mfma_f32_32x32x16_bf16, 32 cycles each):
//   * workgroup = 4 waves = 128 batch rows; wave w owns rows [32w, 32w+32).
//   * A operand (activations): each wave keeps its 32 x K slab in REGISTERS for the whole kernel -- K/16 fragments of
//     8 bf16 (lane (r, hf) holds row r, k = 16 s + 8 hf .. +7), loaded once straight from HBM; the episode-start mask
//     (keep_b) is applied to the h part of the slab as it is loaded.
//   * B operand (weights): streamed through LDS in chunks of 64 weight rows (a gate PAIR x 32 hidden units) x KC
//     columns, rows padded by 16 B so a ds_read_b128 of 16 consecutive rows touches all 64 banks once (measured:
//     SQ_LDS_BANK_CONFLICT = 0).  Every wave reads the same chunk: one global read per workgroup, 4x LDS reuse.  Chunks
//     rotate through three LDS buffers filled by LDS-DMA two chunks ahead of the MFMAs.
//     Two workgroups are resident per CU (75 KB LDS, <= 256 VGPRs each).
//   * per hidden slice (32 units) a wave makes two passes over K: gates (i, g) -> sigmoid(i) tanh(g), then (f, o).
//     Lane (c, hf) owns hidden unit c of the slice for 16 batch rows in every 32x32 accumulator, so the cell update is
//     purely lane-local; h' (bf16) and c' (fp32) go straight from registers to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include "../../include/fdyn.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int BM = 128;            // batch rows per workgroup
constexpr int NSLICE = 32;         // hidden units per slice
constexpr int LDS_PAD = 8;         // bf16 elements (16 B) of row padding

// explicit address spaces for the LDS-DMA builtin (the low 32 bits of a generic LDS address are the LDS offset)
typedef const __attribute__((address_space(1))) void* global_cptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ global_cptr_t as_global(const void* p) { return reinterpret_cast<global_cptr_t>(reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ lds_ptr_t as_lds(void* p) { return reinterpret_cast<lds_ptr_t>(reinterpret_cast<uintptr_t>(p)); }

// 64 lanes x 16 bytes from global memory to 1 KB of LDS starting at `l` (wave-uniform), no registers in between.
// Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: the compiler cannot tell which of the three LDS
// buffers a DMA targets and puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY of them, which serialises the
// prefetch (seen in the ISA).  Hidden from its bookkeeping, the DMA only ever makes the compiler's own vmcnt waits more
// conservative (loads retire in order); the waits that order DMA against LDS reads are written by hand below.
__device__ __forceinline__ void dma16(const void* g, void* l)
{
    const uint32_t lds_off = uint32_t(reinterpret_cast<uintptr_t>(l));      // low 32 bits of a generic LDS address = LDS offset
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_off), "v"(g) : "memory");      // m0 is reserved: the compiler never keeps a live value in it here
}
__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
// the same two functions of (acc + bias) with the bias folded into the exponent's FMA: sb = -bias * log2(e), tb = 2 bias log2(e)
__device__ __forceinline__ float sigmoid_b(float acc, float sb) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc, -1.4426950408889634f, sb))); }
__device__ __forceinline__ float tanh_b(float acc, float tb) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(__builtin_fmaf(acc, 2.8853900817779268f, tb)) + 1.0f), 1.0f); }
__device__ __forceinline__ uint16_t f2bf(float f)
{   // round-to-nearest-even, NaN stays NaN: one v_cvt_pk_bf16_f32 on gfx950 (the shift/add/select form costs six VALU ops)
    return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: a jump over immediates (the instruction takes no register).  Granularity 4,
// rounded DOWN (waits for at most 3 operations more than asked).  The waits of this kernel are COUNTED: vmcnt retires in
// issue order and loads, LDS-DMA and stores share the one counter, so "the DMA has landed" written as a small constant would
// also wait for every store the wave issued after that DMA.
__device__ __forceinline__ void wait_vmcnt(int n)
{
    n = n > 60 ? 60 : n;
    switch (n >> 2) {
#define FD_W(I) case I: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * I) : "memory"); break;
        FD_W(0) FD_W(1) FD_W(2) FD_W(3) FD_W(4) FD_W(5) FD_W(6) FD_W(7) FD_W(8) FD_W(9) FD_W(10) FD_W(11) FD_W(12) FD_W(13) FD_W(14)
        default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
#undef FD_W
    }
}

// Scheduling pattern for a block that holds N MFMAs and, emitted after them in the source, an epilogue share: after each
// MFMA (32 cycles of the matrix pipe, 8 of the issue port) one LDS operand read for a later MFMA, NV VALU / transcendental
// instructions and NST stores.  (The builtin takes constants only: hence the recursion.)
template <int N, int NV, int NST>
__device__ __forceinline__ void sched_mfma_valu()
{
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        if constexpr (NST > 0) __builtin_amdgcn_sched_group_barrier(0x040, NST, 0);
        sched_mfma_valu<N - 1, NV, NST>();
    }
}

// KX = input width, KH = recurrent width (0 = zero-state layer: no h input, no c_prev, no forget gate).
//
// Register budget is what shapes this kernel (256 VGPRs at two waves per SIMD): the activation slab is 96, so the four
// gates of a slice are produced as two PAIRS ("units") -- (i, g) first, folded to sigmoid(i)*tanh(g) (16 registers), then
// (f, o).
//
// Round 2: the point-wise epilogue of a unit is SOFTWARE-PIPELINED under the MFMAs of the next unit, in one instruction
// stream.  Round 1 ran [MFMAs of a unit] -> [its epilogue] back to back in every wave; the barriers kept all waves of a
// workgroup -- and in practice both workgroups of a CU -- in the same phase, so ~36 us of matrix work and ~40 us of
// transcendental / VALU / memory work ADDED (MFMA pipe 26-30 % busy).  Re-arranging WAVES did not help (two
// halves of a 512-thread workgroup in anti-phase were slower).  Here each wave keeps two accumulator sets: while unit u + 1
// multiplies into one, the epilogue of unit u reads the other, a few elements per k-step, in the issue slots the matrix pipe
// leaves free (an MFMA occupies the pipe for 32 cycles and the issue port for 8).
//   block group P0 of slice s : MFMAs of (s, i/g) -> set A   ||  epilogue of (s-1, f/o): c', h' of slice s-1, stores
//   block group P1 of slice s : MFMAs of (s, f/o) -> set B   ||  epilogue of (s, i/g): ig = sigmoid(i) tanh(g)
// The cell state a slice needs is requested at the top of its P1 group, a whole unit before its use.
//
// The weight stream moves by LDS-DMA (global_load_lds_dwordx4, new on gfx950): a chunk goes from L2 straight into LDS
// without passing through registers, three LDS buffers in rotation, chunk q + 2 requested while chunk q is multiplied.
// DMA layout rule: the 64 lanes of one instruction write 64 consecutive 16-byte slots of LDS; the padded chunk (64 rows x
// (KC + 8) bf16) is therefore moved as 64 * (KC/8 + 1) slots, the pad slot of each row re-reading the row's last vector.
// Wave w issues slot groups w, w + 4, ... -- every wave the SAME number NG (the surplus re-issues a group: a duplicate write of
// the same bytes), so that each wait can say exactly how many younger operations may stay in flight.
template <int KX, int KH, bool TRAIN>
__global__ void __launch_bounds__(256, 2)
lstm_cell_mfma_dma_kernel(const uint16_t* __restrict__ x /*[B][KX] bf16*/, const uint16_t* __restrict__ h_prev /*[B][KH] bf16*/,
                      const float* __restrict__ c_prev /*[B][H]*/, const float* __restrict__ keep /*[B] or null*/,
                      const uint16_t* __restrict__ W /*[4H][KX+KH] bf16*/, const float* __restrict__ bias /*[4H]*/,
                      uint16_t* __restrict__ h_out /*[B][H] bf16*/, float* __restrict__ c_out /*[B][H]*/,
                      float* __restrict__ h_out_f32 /*[B][H] or null*/, int64_t B, int H, int split,
                      // BPTT forward (all null / 0 in the rollout): the activated gates for the backward pass, and h' * keep_next
                      // packed into the recurrent columns of the next step's [x | h] input row
                      uint16_t* __restrict__ act_out = nullptr /*[B][act_gates * H] bf16: sigmoid(i), [sigmoid(f),] tanh(g), sigmoid(o)*/,
                      int act_gates = 4 /*4: (i, f, g, o); 3: (i, g, o), zero-state layers*/,
                      uint16_t* __restrict__ h_next = nullptr /*row stride next_stride elements*/, int64_t next_stride = 0,
                      const float* __restrict__ keep_next = nullptr /*[B] or null*/)
{
    constexpr int K = KX + KH;
    constexpr int KSTEPS = K / 16;
    constexpr int NCHUNK = (K + 191) / 192;               // K-chunks per unit: 384 -> 2 x 192, 256 -> 2 x 128, 128 -> 1
    constexpr int KC = K / NCHUNK;
    constexpr int KC_STEPS = KC / 16;
    constexpr int ROW = KC + LDS_PAD;                     // padded LDS row (bf16 elements)
    constexpr bool RECUR = KH > 0;
    constexpr int THREADS = 256;
    constexpr int CROWS = 2 * NSLICE;                     // weight rows per chunk: one gate pair x 32 hidden units
    constexpr int SLOTS_PER_ROW = ROW / 8;                // 16-byte slots per padded row: KC/8 data + 1 pad
    constexpr int TOTAL_SLOTS = CROWS * SLOTS_PER_ROW;    // 1600 (KC = 192) or 1088 (KC = 128): multiples of 64
    constexpr int NGROUPS = TOTAL_SLOTS / 64;             // 64-slot groups per chunk (25 / 17)
    constexpr int NG = (NGROUPS + 3) / 4;                 // DMA instructions per chunk per wave (7 / 5), the same for every wave
    static_assert(TOTAL_SLOTS % 64 == 0 && NG <= 7, "chunk must be a whole number of 64-slot groups");
    constexpr int BUF = CROWS * ROW;
    constexpr int EPC = 16 / NCHUNK;                      // epilogue elements folded into one chunk's MFMAs
    constexpr int PD = 2;                                 // k-steps a B-operand LDS read runs ahead of its MFMAs
    __shared__ __attribute__((aligned(16))) uint16_t s_w[3 * BUF];
    __shared__ float s_keep[BM];
    __shared__ float s_keepn[BM];                         // keep_next of the block's rows (BPTT forward)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int64_t row0 = int64_t(blockIdx.x) * BM + wave * 32;
    const int64_t my_row = row0 + r;                      // A-operand row of this lane
    const bool row_ok = my_row < B;
    // `split` workgroups share one 128-row block, each taking n_slices / split of the hidden slices (blockIdx.y): small
    // batches (B / 128 < 2 x CUs) would otherwise leave most of the chip idle.  Costs one extra read of the row block's
    // activation slab per split.
    const int n_slices = (H / NSLICE) / split;
    const int slice0 = int(blockIdx.y) * n_slices;

    // chunk (sl, pass, ch): 64 weight rows = gates (pass, pass + 2) of hidden units sl*32 .. +31, columns ch*KC .. +KC;
    // pass 0 = (i, g), pass 1 = (f, o).  Zero-state layers have no use for f: their pass 1 still moves (f, o) -- every
    // address stays inside W -- but only multiplies the o half.
    // Slot v of a chunk = padded row v / SLOTS_PER_ROW, 16-byte column v % SLOTS_PER_ROW (the pad column re-reads the last
    // data column).  Lane offsets are per-thread constants: group g_ = (uwave + 4 jj) mod NGROUPS, slot v = 64 g_ + lane.
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
#define FD_GRP(JJ) ((uwave + 4 * (JJ)) % NGROUPS)
#define FD_SLOT(JJ) (FD_GRP(JJ) * 64 + lane)
#define FD_SOFF(JJ) ((((FD_SLOT(JJ) / SLOTS_PER_ROW) >> 5) * 2 * H + ((FD_SLOT(JJ) / SLOTS_PER_ROW) & 31)) * K + \
                     ((FD_SLOT(JJ) % SLOTS_PER_ROW) < SLOTS_PER_ROW - 1 ? (FD_SLOT(JJ) % SLOTS_PER_ROW) : SLOTS_PER_ROW - 2) * 8)
    const int soff0 = FD_SOFF(0), soff1 = FD_SOFF(1), soff2 = FD_SOFF(2), soff3 = FD_SOFF(3), soff4 = FD_SOFF(4), soff5 = FD_SOFF(5),
              soff6 = FD_SOFF(6);
    (void)soff5; (void)soff6;
#define FD_ORIGIN(SL, PASS, CH) (W + (int64_t(PASS) * H + (SL) * NSLICE) * K + (CH) * KC)
#define FD_D1(JJ, ORG, BUFI) if constexpr (NG > JJ) dma16((ORG) + soff##JJ, s_w + (BUFI) * BUF + FD_GRP(JJ) * 512);
#define FD_DMA(SL, PASS, CH, BUFI) { const uint16_t* org_ = FD_ORIGIN(SL, PASS, CH); FD_D1(0, org_, BUFI) FD_D1(1, org_, BUFI) \
        FD_D1(2, org_, BUFI) FD_D1(3, org_, BUFI) FD_D1(4, org_, BUFI) FD_D1(5, org_, BUFI) FD_D1(6, org_, BUFI) }

    // Workgroups walk the hidden slices in ROTATED order.  In natural order all 512 workgroups stream the same 25 KB of W
    // at the same moment: in each XCD's L2 one hot line is served to 64 workgroups while the other channels idle.  Blocks b and
    // b + 8 share an XCD (round-robin dispatch), so the rotation mixes b and b / 8.  Speed only -- any placement gives the
    // same result.
    const int sl_start = int((blockIdx.x + (blockIdx.x >> 3)) % unsigned(n_slices));
#define FD_SL(I) (slice0 + ((I) + sl_start) % n_slices)
    constexpr int CHUNKS_PER_SLICE = 2 * NCHUNK;
    // chunk q of this workgroup's sequence: slice q / CHUNKS_PER_SLICE, pass (q / NCHUNK) & 1, column block q % NCHUNK
#define FD_DMA_SEQ(Q, BUFI) FD_DMA(FD_SL((Q) / CHUNKS_PER_SLICE), ((Q) / NCHUNK) & 1, (Q) % NCHUNK, BUFI)
    const int total_chunks = n_slices * CHUNKS_PER_SLICE;
    FD_DMA_SEQ(0, 0)                                      // start the weight stream before the activation slab
    if (total_chunks > 1) { FD_DMA_SEQ(1, 1) }

    if (RECUR) {                                          // episode-start mask -> LDS (before the slab loads: its wait must not cover them)
        for (int i = tid; i < BM; i += THREADS) {
            const int64_t b = int64_t(blockIdx.x) * BM + i;
            s_keep[i] = (keep && b < B) ? keep[b] : 1.0f;
        }
    }
    if (TRAIN && h_next) {
        for (int i = tid; i < BM; i += THREADS) {
            const int64_t b = int64_t(blockIdx.x) * BM + i;
            s_keepn[i] = (keep_next && b < B) ? keep_next[b] : 1.0f;
        }
    }
    // ---- A slab -> registers (masked h part).  Branch-free: rows past B read row B-1 (their results are never stored) and
    // the episode-start mask is a select, so all K/16 loads are in flight together (a guarded load per k-step makes
    // hipcc branch and drain vmcnt around each one).
    bf16x8_t a[KSTEPS];
    {
        const int64_t lrow = row_ok ? my_row : (B - 1);
        const float kp = (RECUR && keep) ? keep[lrow] : 1.0f;
        const uint16_t* xr = x + lrow * KX;
        const uint16_t* hr = RECUR ? h_prev + lrow * KH : x;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int k = 16 * s + 8 * hf;
            uint4 v;
            if (16 * s < KX) {
                v = *reinterpret_cast<const uint4*>(xr + k);
            } else {
                v = *reinterpret_cast<const uint4*>(hr + (k - KX));
                v = (kp != 0.0f) ? v : make_uint4(0, 0, 0, 0);
            }
            a[s] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    // chunks 0 and 1 have landed once no more than the KSTEPS slab loads (all younger) are in flight: the slab itself keeps
    // arriving under the first MFMAs (the compiler waits per fragment); a vmcnt(0) here held every wave until its whole 24 KB
    // slab was in registers
    wait_vmcnt(KSTEPS);
    __syncthreads();

    // Addresses of the epilogue: [per-wave base pointer] + [one 32-bit per-lane offset] + [compile-time row constant * H].
    // Written as b * H + col per element, LLVM hoists sixteen 64-bit row addresses per output array out of the slice loop and
    // the activation slab spills.  Uniform base pointers + UNSIGNED 32-bit BYTE offsets (a 32-row tile spans < 64 KB): the
    // loads / stores take the "scalar base + vector offset" form.  (row0 recomputed from the readfirstlane'd wave index: the
    // compiler then knows the bases are wave-uniform -> SGPRs)
    const bool full = row0 + 32 <= B;                     // wave-uniform: every row of this wave's tile exists
    const int rows_left = row0 < B ? int(B - row0 < 32 ? B - row0 : 32) : 0;   // rows of this tile that exist
    const int64_t urow0 = int64_t(blockIdx.x) * BM + uwave * 32;
    const int64_t uwave_off = (urow0 < B ? urow0 : 0) * H;       // a wave wholly past B reads row 0, stores nothing
    const float* cp_base = RECUR ? c_prev + uwave_off : nullptr;
    float* c_base = c_out ? c_out + uwave_off : nullptr;
    float* h32_base = h_out_f32 ? h_out_f32 + uwave_off : nullptr;
    uint16_t* h_base = h_out + uwave_off;
    const uint32_t aH = uint32_t(act_gates * H);          // activated-gate row length
    // TRAIN is a separate instantiation: the rollout's kernel must not pay registers for these (it sits at the 256-register edge)
    uint16_t* act_base = (TRAIN && act_out) ? act_out + (urow0 < B ? urow0 : 0) * int64_t(aH) : nullptr;
    uint16_t* hn_base = (TRAIN && h_next) ? h_next + (urow0 < B ? urow0 : 0) * next_stride : nullptr;
    const uint32_t aoff = uint32_t(hf * 4) * aH, noff = uint32_t(hf * 4) * uint32_t(next_stride);
    const uint32_t ag_f = uint32_t(H), ag_g = uint32_t((act_gates - 2) * H), ag_o = uint32_t((act_gates - 1) * H);     // gate i at 0
    const uint32_t uoff = uint32_t(hf * 4 * H), uH = uint32_t(H);          // + the slice's hidden unit per use
    auto ld_f32 = [](const float* b, uint32_t boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + boff); };
    auto st_f32 = [](float* b, uint32_t boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(b) + boff) = v; };
    auto st_u16 = [](uint16_t* b, uint32_t boff, uint16_t v) { *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(b) + boff) = v; };
    // store instructions one (f, o) epilogue element issues
    const int st_per_elem = 1 + (c_base != nullptr) + (h32_base != nullptr) + (act_base ? (RECUR ? 2 : 1) : 0) + (hn_base != nullptr);

    // ---- VMEM bookkeeping for the counted waits (all wave-uniform)
    int after_last_dma = 0;    // operations issued after the most recent chunk request
    int after_prev_dma = 0;    // ... after the request BEFORE it (the chunk the next barrier must have)
    int after_cp = 0;          // ... after the pending slice's c_prev loads
#define FD_ISSUED(N) { after_last_dma += (N); after_prev_dma += (N); after_cp += (N); }

    f32x16_t accA0, accA1, accB0, accB1;
    float ig[16], cp[16];
    uint32_t pcol = 0;         // hidden unit of the slice whose (f, o) epilogue is pending
    float pbo = 0.0f, pbf = 0.0f;

    // epilogue of the pending slice's (f, o) unit, element e: c' = ig + sigmoid(f) keep c ; h' = sigmoid(o) tanh(c') ; stores.
    // FAST: full 32-row tile, c' wanted, no fp32 copy of h' -- every condition is compile-time, so the element is straight-line
    // code the scheduler can place between the MFMAs it is emitted next to.
    auto epi1 = [&](int e, auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
        float c = ig[e];
        const float sf = RECUR ? sigmoid_b(accB0[e], pbf) : 0.0f;
        if (RECUR) c += sf * (s_keep[wave * 32 + lr] * cp[e]);
        const float so = sigmoid_b(accB1[e], pbo);
        const float hv = so * tanh_(c);
        const uint32_t off = uoff + pcol + uint32_t((e & 3) + 8 * (e >> 2)) * uH;
        if constexpr (FAST) {
            st_f32(c_base, off * 4u, c);
            st_u16(h_base, off * 2u, f2bf(hv));
        } else if (full || lr < rows_left) {
            if (c_base) st_f32(c_base, off * 4u, c);
            st_u16(h_base, off * 2u, f2bf(hv));
            if (h32_base) st_f32(h32_base, off * 4u, hv);
            if (TRAIN && act_base) {
                const uint32_t ao = aoff + pcol + uint32_t((e & 3) + 8 * (e >> 2)) * aH;
                if (RECUR) st_u16(act_base, (ao + ag_f) * 2u, f2bf(sf));
                st_u16(act_base, (ao + ag_o) * 2u, f2bf(so));
            }
            if (TRAIN && hn_base)
                st_u16(hn_base, (noff + pcol + uint32_t((e & 3) + 8 * (e >> 2)) * uint32_t(next_stride)) * 2u,
                       f2bf(hv * s_keepn[wave * 32 + lr]));
        }
    };

    int q = 0;                                            // chunk counter; buffers rotate cur -> nx1 -> nx2
    int cur = 0, nx1 = 1, nx2 = 2;
    // one chunk's block: request chunk q + 2, multiply chunk q, fold a share of the pending epilogue in, make chunk q + 1 visible
    /* the request is UNCONDITIONAL (past the last chunk it re-requests a chunk of the last slice into the free buffer: a     \
       harmless duplicate), so every count below is a compile-time constant and every wait a single instruction */               \
#define FD_BLOCK_BEGIN(SI, LIN2)                                                                                       \
        {                                                                                                              \
            const int si2_ = (SI) + (LIN2) / CHUNKS_PER_SLICE;                                                         \
            FD_DMA(FD_SL(si2_ < n_slices ? si2_ : n_slices - 1), ((LIN2) % CHUNKS_PER_SLICE) / NCHUNK, (LIN2) % NCHUNK, nx2) \
            after_prev_dma = after_last_dma + NG; after_last_dma = 0; after_cp += NG;                                  \
        }
#define FD_BLOCK_END                                                                                                   \
        wait_vmcnt(after_prev_dma);                   /* chunk q + 1 is in LDS; everything younger may stay in flight */ \
        __syncthreads();                              /* ... and everyone's part; buffer `cur` is free for chunk q + 3 */ \
        { const int t_ = cur; cur = nx1; nx1 = nx2; nx2 = t_; }                                                        \
        ++q;

    // one hidden slice: PENDING = a previous slice's (f, o) epilogue is folded into this slice's first MFMA group
    auto slice = [&](int si, auto pending_c, auto fast_c) {
        constexpr bool PENDING = decltype(pending_c)::value;
        const int sl = FD_SL(si);
        const uint32_t col = uint32_t(sl * NSLICE + r);
        // ================= P0: (i, g) of slice sl -> set A, under it the (f, o) epilogue of the previous slice =================
#pragma unroll
        for (int e = 0; e < 16; ++e) { accA0[e] = 0.0f; accA1[e] = 0.0f; }
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            FD_BLOCK_BEGIN(si, ch + 2)
            if (ch == 0 && PENDING && RECUR) wait_vmcnt(after_cp);          // the pending slice's c_prev has arrived
            const uint16_t* wb = s_w + cur * BUF;
            // B operands are read PD k-steps ahead of the MFMAs that use them.  Left to the compiler every ds_read_b128 landed
            // in the one register quad its MFMA then consumed -- read, s_waitcnt lgkmcnt(0), MFMA, read, wait, MFMA -- so each
            // MFMA paid the full LDS latency (~100+ cycles with eight waves reading) on a 32-cycle instruction: the reason the
            // matrix pipe sat at 26 % however the rest of the kernel was arranged.
            uint4 pb0[PD + 1], pb1[PD + 1];
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                pb0[j] = *reinterpret_cast<const uint4*>(wb + r * ROW + j * 16 + hf * 8);
                pb1[j] = *reinterpret_cast<const uint4*>(wb + (NSLICE + r) * ROW + j * 16 + hf * 8);
            }
#pragma unroll
            for (int ks = 0; ks < KC_STEPS; ++ks) {
                const bf16x8_t af = a[ch * KC_STEPS + ks];
                if (ks + PD < KC_STEPS) {
                    pb0[(ks + PD) % (PD + 1)] = *reinterpret_cast<const uint4*>(wb + r * ROW + (ks + PD) * 16 + hf * 8);
                    pb1[(ks + PD) % (PD + 1)] = *reinterpret_cast<const uint4*>(wb + (NSLICE + r) * ROW + (ks + PD) * 16 + hf * 8);
                }
                accA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, pb0[ks % (PD + 1)]), accA0, 0, 0, 0);
                accA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, pb1[ks % (PD + 1)]), accA1, 0, 0, 0);
            }
            if constexpr (PENDING) {
#pragma unroll
                for (int e = ch * EPC; e < (ch + 1) * EPC; ++e) epi1(e, fast_c);
                if constexpr (decltype(fast_c)::value) {
                    FD_ISSUED(EPC * 2)
                    // interleave: after each MFMA (32 cycles of the matrix pipe, 8 of the issue port) its LDS operand read
                    // for the next one and a share of the epilogue's VALU / transcendental / store instructions
                    sched_mfma_valu<2 * KC_STEPS, (EPC * 20 + 2 * KC_STEPS - 1) / (2 * KC_STEPS), 1>();
                } else if (full || rows_left > 0) {
                    FD_ISSUED(EPC * st_per_elem)
                }
            }
            FD_BLOCK_END
        }
        // ================= P1: (f, o) of slice sl -> set B, under it the (i, g) epilogue of this slice =================
        if (RECUR) {                                          // this slice's cell state: requested a whole unit before its use
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
                const uint32_t off = (full || lr < rows_left) ? uoff + col + uint32_t((e & 3) + 8 * (e >> 2)) * uH : col;
                cp[e] = ld_f32(cp_base, off * 4u);
            }
            FD_ISSUED(16)
            after_cp = 0;
        }
        const float bi = bias[col] * -1.4426950408889634f, bg = bias[2 * H + col] * 2.8853900817779268f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { accB0[e] = 0.0f; accB1[e] = 0.0f; }
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ++ch) {
            FD_BLOCK_BEGIN(si, NCHUNK + ch + 2)
            const uint16_t* wb = s_w + cur * BUF;
            uint4 pb0[PD + 1], pb1[PD + 1];
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                if (RECUR) pb0[j] = *reinterpret_cast<const uint4*>(wb + r * ROW + j * 16 + hf * 8);
                pb1[j] = *reinterpret_cast<const uint4*>(wb + (NSLICE + r) * ROW + j * 16 + hf * 8);
            }
#pragma unroll
            for (int ks = 0; ks < KC_STEPS; ++ks) {
                const bf16x8_t af = a[ch * KC_STEPS + ks];
                if (ks + PD < KC_STEPS) {
                    if (RECUR) pb0[(ks + PD) % (PD + 1)] = *reinterpret_cast<const uint4*>(wb + r * ROW + (ks + PD) * 16 + hf * 8);
                    pb1[(ks + PD) % (PD + 1)] = *reinterpret_cast<const uint4*>(wb + (NSLICE + r) * ROW + (ks + PD) * 16 + hf * 8);
                }
                if (RECUR)                                    // rows 0..31 of the chunk: gate f (zero-state layers have none)
                    accB0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, pb0[ks % (PD + 1)]), accB0, 0, 0, 0);
                accB1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, pb1[ks % (PD + 1)]), accB1, 0, 0, 0);   // o
            }
            // C/D map of a 32x32 tile: col = lane & 31 (hidden unit), row = (e & 3) + 8 (e >> 2) + 4 hf
#pragma unroll
            for (int e = ch * EPC; e < (ch + 1) * EPC; ++e) {
                const float si = sigmoid_b(accA0[e], bi), tg = tanh_b(accA1[e], bg);
                ig[e] = si * tg;
                if constexpr (TRAIN && !decltype(fast_c)::value) {
                    const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
                    if (act_base && (full || lr < rows_left)) {
                        const uint32_t ao = aoff + col + uint32_t((e & 3) + 8 * (e >> 2)) * aH;
                        st_u16(act_base, ao * 2u, f2bf(si));
                        st_u16(act_base, (ao + ag_g) * 2u, f2bf(tg));
                    }
                }
            }
            if constexpr (decltype(fast_c)::value) {
                sched_mfma_valu<(RECUR ? 2 : 1) * KC_STEPS, (EPC * 11 + 2 * KC_STEPS - 1) / (2 * KC_STEPS) + (RECUR ? 0 : 4), 0>();
            } else if (TRAIN && act_base && (full || rows_left > 0)) {
                FD_ISSUED(EPC * 2)
            }
            FD_BLOCK_END
        }
        pcol = col;
        pbo = bias[3 * H + col] * -1.4426950408889634f;
        pbf = RECUR ? bias[H + col] * -1.4426950408889634f : 0.0f;
    };

    using T_ = std::true_type; using F_ = std::false_type;
    const bool fast = !TRAIN && full && c_base != nullptr && h32_base == nullptr;     // wave-uniform
    if (fast) {
        slice(0, F_{}, T_{});
        for (int si = 1; si < n_slices; ++si) slice(si, T_{}, T_{});
    } else {
        slice(0, F_{}, F_{});
        for (int si = 1; si < n_slices; ++si) slice(si, T_{}, F_{});
    }
    // ---- drain: the last slice's (f, o) epilogue has nothing to hide under
    if (RECUR) wait_vmcnt(after_cp);
#pragma unroll
    for (int e = 0; e < 16; ++e) epi1(e, F_{});
}

template <int KX, int KH>
int launch(const void* x, const void* h_prev, const float* c_prev, const float* keep, const void* W, const float* bias,
           void* h_out, float* c_out, float* h_out_f32, int64_t B, int H, hipStream_t st, void* act_out = nullptr, int act_gates = 4,
           void* h_next = nullptr, int64_t next_stride = 0, const float* keep_next = nullptr)
{
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    const int64_t row_blocks = (B + BM - 1) / BM;
    int split = 1;                                        // aim for >= 2 workgroups per CU, split a power of two <= H/32
    while (split < H / NSLICE && row_blocks * split < int64_t(2) * cus) split *= 2;
    while ((H / NSLICE) % split) split /= 2;
    if (act_out || h_next)
        hipLaunchKernelGGL((lstm_cell_mfma_dma_kernel<KX, KH, true>), dim3(unsigned(row_blocks), unsigned(split)), dim3(256), 0, st,
                           (const uint16_t*)x, (const uint16_t*)h_prev, c_prev, keep, (const uint16_t*)W, bias, (uint16_t*)h_out, c_out,
                           h_out_f32, B, H, split, (uint16_t*)act_out, act_gates, (uint16_t*)h_next, next_stride, keep_next);
    else
        hipLaunchKernelGGL((lstm_cell_mfma_dma_kernel<KX, KH, false>), dim3(unsigned(row_blocks), unsigned(split)), dim3(256), 0, st,
                           (const uint16_t*)x, (const uint16_t*)h_prev, c_prev, keep, (const uint16_t*)W, bias, (uint16_t*)h_out, c_out,
                           h_out_f32, B, H, split, (uint16_t*)nullptr, 4, (uint16_t*)nullptr, int64_t(0), (const float*)nullptr);
    return int(hipGetLastError());
}

}  // namespace

extern "C" int fdyn_lstm_cell_mfma64_try(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                                         const void* W, const float* bias, void* h_out, float* c_out, int64_t B, int H, void* stream);   // lstm_mfma64.hip

static bool use_mfma64() { static const bool u = [] { const char* e = getenv("FDYN_MFMA64"); return !(e && e[0] == '0'); }(); return u; }

// 1 = fdyn_lstm_cell_mfma may be called with h_out == h_prev and c_out == c_prev for this shape: the one-wave-per-SIMD kernel
// (lstm_mfma64.hip) gives every wave 64 whole rows -- it loads their h before it stores any h' and reads each c_prev word in
// the lane that then overwrites it.  The tiled kernel below splits a row's hidden slices over workgroups: never in place.
extern "C" int fdyn_lstm_cell_mfma_inplace_ok(int kx, int kh, int H, int64_t B)
{
    return use_mfma64() && kx == 128 && kh == 256 && H == 256 && B >= 128 * 256 && B % 256 == 0;
}

extern "C" int fdyn_lstm_cell_mfma(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                                   const void* W, const float* bias, void* h_out, float* c_out, float* h_out_f32,
                                   int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % NSLICE) return FDYN_ERR_BAD_SIZE;
    if (!x || !W || !bias || !h_out) return FDYN_ERR_NULL;
    if (kh > 0 && (!h_prev || !c_prev)) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    // a state updated in place is only safe in the kernel that owns whole rows per wave; anywhere else it would corrupt silently
    if (kh > 0 && (h_out == h_prev || (c_out && c_out == c_prev)) && !(fdyn_lstm_cell_mfma_inplace_ok(kx, kh, H, B) && !h_out_f32 && c_out))
        return FDYN_ERR_BAD_SIZE;
    hipStream_t st = (hipStream_t)stream;
    // whole 256-row blocks of the (128, 256) cell, enough of them to give every CU one: the one-wave-per-SIMD kernel
    const bool use64 = use_mfma64();
    if (use64 && !h_out_f32 && B >= 128 * 256) {
        const int rc = fdyn_lstm_cell_mfma64_try(x, kx, h_prev, kh, c_prev, keep, W, bias, h_out, c_out, B, H, stream);
        if (rc) return rc > 0 ? FDYN_OK : int(hipGetLastError());
    }
    if (kx == 128 && kh == 256) return launch<128, 256>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 128 && kh == 0) return launch<128, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 256 && kh == 0) return launch<256, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 128 && kh == 128) return launch<128, 128>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    return FDYN_ERR_BAD_SIZE;      // supported (input, recurrent) widths: (128,256) (128,0) (256,0) (128,128)
}

extern "C" int fdyn_lstm_cell_mfma64_pair_try(const void* x, int kx, const float* keep, int kh, int64_t B, int H,
                                              const void* h_prev0, const float* c_prev0, const void* W0, const float* bias0, void* h_out0, float* c_out0,
                                              const void* h_prev1, const float* c_prev1, const void* W1, const float* bias1, void* h_out1, float* c_out1,
                                              void* stream);   // lstm_mfma64.hip

// Two cells of the same shape on the same input and keep mask -- sb3_contrib's MlpLstmPolicy keeps an actor and a critic LSTM
// (learned_controllers/networks/lstm_policy.py:107-136) -- in one launch where the one-wave-per-SIMD kernel applies, otherwise as
// two calls of fdyn_lstm_cell_mfma.  Same operand rules (aliased state only where fdyn_lstm_cell_mfma_inplace_ok says so).
extern "C" int fdyn_lstm_cell_mfma_pair(const void* x, int kx, const float* keep, int kh, int64_t B, int H,
                                        const void* h_prev0, const float* c_prev0, const void* W0, const float* bias0, void* h_out0, float* c_out0,
                                        const void* h_prev1, const float* c_prev1, const void* W1, const float* bias1, void* h_out1, float* c_out1,
                                        void* stream)
{
    if (B < 0 || H <= 0 || H % NSLICE || kh <= 0) return FDYN_ERR_BAD_SIZE;
    if (!x || !W0 || !bias0 || !h_out0 || !h_prev0 || !c_prev0 || !W1 || !bias1 || !h_out1 || !h_prev1 || !c_prev1) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const bool aliased = h_out0 == h_prev0 || (c_out0 && c_out0 == c_prev0) || h_out1 == h_prev1 || (c_out1 && c_out1 == c_prev1);
    if (aliased && !(fdyn_lstm_cell_mfma_inplace_ok(kx, kh, H, B) && c_out0 && c_out1)) return FDYN_ERR_BAD_SIZE;
    if (use_mfma64() && B >= 128 * 256) {
        const int rc = fdyn_lstm_cell_mfma64_pair_try(x, kx, keep, kh, B, H, h_prev0, c_prev0, W0, bias0, h_out0, c_out0,
                                                      h_prev1, c_prev1, W1, bias1, h_out1, c_out1, stream);
        if (rc) return rc > 0 ? FDYN_OK : int(hipGetLastError());
    }
    const int e = fdyn_lstm_cell_mfma(x, kx, h_prev0, kh, c_prev0, keep, W0, bias0, h_out0, c_out0, nullptr, B, H, stream);
    return e != FDYN_OK ? e : fdyn_lstm_cell_mfma(x, kx, h_prev1, kh, c_prev1, keep, W1, bias1, h_out1, c_out1, nullptr, B, H, stream);
}

// BPTT forward of the same cell: besides h' and c' the kernel leaves what the backward pass needs -- the activated gates
// (bf16 [B][act_gates * H]; zero-state layers, kh = 0, use the three-gate layout (i, g, o) and take c_out = NULL) -- and packs
// h' * keep_next into the recurrent columns of the next step's input row.  The [B, 4H] pre-activations never exist in HBM:
// the un-fused training step wrote them from the GEMM and read them back in the point-wise kernel.
extern "C" int fdyn_lstm_cell_mfma_train(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                                         const void* W, const float* bias, void* h_out, float* c_out, void* act_out,
                                         void* h_next, int64_t next_stride, const float* keep_next, int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % NSLICE || (h_next && next_stride < H)) return FDYN_ERR_BAD_SIZE;
    if (!x || !W || !bias || !h_out || !act_out) return FDYN_ERR_NULL;
    if (kh > 0 && (!h_prev || !c_prev || !c_out)) return FDYN_ERR_NULL;
    if (int64_t(32) * 4 * H * 2 > (int64_t(1) << 31) || int64_t(32) * next_stride * 2 > (int64_t(1) << 31)) return FDYN_ERR_BAD_SIZE;
    if (B == 0) return FDYN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int ng = kh > 0 ? 4 : 3;
    if (kx == 128 && kh == 256) return launch<128, 256>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, nullptr, B, H, st, act_out, ng, h_next, next_stride, keep_next);
    if (kx == 128 && kh == 0) return launch<128, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, nullptr, B, H, st, act_out, ng, h_next, next_stride, keep_next);
    if (kx == 256 && kh == 0) return launch<256, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, nullptr, B, H, st, act_out, ng, h_next, next_stride, keep_next);
    if (kx == 128 && kh == 128) return launch<128, 128>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, nullptr, B, H, st, act_out, ng, h_next, next_stride, keep_next);
    return FDYN_ERR_BAD_SIZE;
}
