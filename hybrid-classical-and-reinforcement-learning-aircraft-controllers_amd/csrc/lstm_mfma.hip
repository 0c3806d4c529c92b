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
__device__ __forceinline__ void wait_vmcnt_le(int n)       // n in {0, 4, 6}: the values NG_WAVE - 1 takes
{
    if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// KX = input width, KH = recurrent width (0 = zero-state layer: no h input, no c_prev, no forget gate).
//
// Register budget is what shapes this kernel (256 VGPRs at two waves per SIMD): the activation slab is 96, so the four
// gates of a slice are produced as two PAIRS -- (i, g) first, folded to sigmoid(i)*tanh(g) (16 registers), then (f, o) --
// which halves the live accumulators (32).
//
// The weight stream moves by LDS-DMA (global_load_lds_dwordx4, new on gfx950): a chunk goes from L2 straight into LDS
// without passing through registers, so three LDS buffers are kept in rotation and chunk q + 2 is requested while chunk q
// is multiplied.  (A register-staged double-buffered version of this kernel was measured at 98.7 us against 93.5 us and
// removed in round 2: its chunk period was set by the L2 round trip -- MFMA pipe 25 % busy, a third of the wave cycles
// waiting on VMEM.)
// DMA layout rule: the 64 lanes of one instruction write 64 consecutive 16-byte slots of LDS; the padded chunk (64 rows x
// (KC + 8) bf16) is therefore moved as 64 * (KC/8 + 1) slots, the pad slot of each row re-reading the row's last vector.
// Wave w issues slot groups w, w + 4, ...; each wave waits for its own groups (vmcnt) before the chunk barrier.
template <int KX, int KH>
__global__ void __launch_bounds__(256, 2)
lstm_cell_mfma_dma_kernel(const uint16_t* __restrict__ x /*[B][KX] bf16*/, const uint16_t* __restrict__ h_prev /*[B][KH] bf16*/,
                      const float* __restrict__ c_prev /*[B][H]*/, const float* __restrict__ keep /*[B] or null*/,
                      const uint16_t* __restrict__ W /*[4H][KX+KH] bf16*/, const float* __restrict__ bias /*[4H]*/,
                      uint16_t* __restrict__ h_out /*[B][H] bf16*/, float* __restrict__ c_out /*[B][H]*/,
                      float* __restrict__ h_out_f32 /*[B][H] or null*/, int64_t B, int H, int split)
{
    constexpr int K = KX + KH;
    constexpr int KSTEPS = K / 16;
    constexpr int NCHUNK = (K + 191) / 192;               // K-chunks per pass: 384 -> 2 x 192, 256 -> 2 x 128, 128 -> 1
    constexpr int KC = K / NCHUNK;
    constexpr int KC_STEPS = KC / 16;
    constexpr int ROW = KC + LDS_PAD;                     // padded LDS row (bf16 elements)
    constexpr bool RECUR = KH > 0;
    constexpr int THREADS = 256;
    constexpr int CROWS = 2 * NSLICE;                     // weight rows per chunk: one gate pair x 32 hidden units
    constexpr int SLOTS_PER_ROW = ROW / 8;                // 16-byte slots per padded row: KC/8 data + 1 pad
    constexpr int TOTAL_SLOTS = CROWS * SLOTS_PER_ROW;    // 1600 (KC = 192) or 1088 (KC = 128): multiples of 64
    constexpr int NGROUPS = TOTAL_SLOTS / 64;             // DMA instructions per chunk over the whole workgroup (25 / 17)
    constexpr int NG_WAVE = (NGROUPS + 3) / 4;            // ... per wave, at most (7 / 5)
    static_assert(TOTAL_SLOTS % 64 == 0 && NG_WAVE <= 7, "chunk must be a whole number of 64-slot groups");
    constexpr int BUF = CROWS * ROW;
    __shared__ __attribute__((aligned(16))) uint16_t s_w[3 * BUF];
    __shared__ float s_keep[BM];

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
    // data column).  Lane offsets are per-thread constants: group g_ = uwave + 4 jj, slot v = 64 g_ + lane.
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
#define FD_SLOT(JJ) (((uwave + 4 * (JJ)) * 64 + lane) < TOTAL_SLOTS ? ((uwave + 4 * (JJ)) * 64 + lane) : 0)
#define FD_SOFF(JJ) ((((FD_SLOT(JJ) / SLOTS_PER_ROW) >> 5) * 2 * H + ((FD_SLOT(JJ) / SLOTS_PER_ROW) & 31)) * K + \
                     ((FD_SLOT(JJ) % SLOTS_PER_ROW) < SLOTS_PER_ROW - 1 ? (FD_SLOT(JJ) % SLOTS_PER_ROW) : SLOTS_PER_ROW - 2) * 8)
    const int soff0 = FD_SOFF(0), soff1 = FD_SOFF(1), soff2 = FD_SOFF(2), soff3 = FD_SOFF(3), soff4 = FD_SOFF(4), soff5 = FD_SOFF(5),
              soff6 = FD_SOFF(6);
    (void)soff5; (void)soff6;
#define FD_ORIGIN(SL, PASS, CH) (W + (int64_t(PASS) * H + (SL) * NSLICE) * K + (CH) * KC)
#define FD_D1(JJ, ORG, BUFI) if constexpr (NG_WAVE > JJ) { if (uwave + 4 * (JJ) < NGROUPS) \
        dma16((ORG) + soff##JJ, s_w + (BUFI) * BUF + (uwave + 4 * (JJ)) * 512); }
#define FD_DMA(SL, PASS, CH, BUFI) { const uint16_t* org_ = FD_ORIGIN(SL, PASS, CH); FD_D1(0, org_, BUFI) FD_D1(1, org_, BUFI) \
        FD_D1(2, org_, BUFI) FD_D1(3, org_, BUFI) FD_D1(4, org_, BUFI) FD_D1(5, org_, BUFI) FD_D1(6, org_, BUFI) }
    // every wave has at most NG_WAVE and at least NG_WAVE - 1 groups per chunk; loads retire in order, so "no more than
    // NG_WAVE - 1 outstanding" means everything older than the newest chunk's groups has landed
    // (when no newer chunk was requested, everything must have landed)
#define FD_WAIT_PREV_CHUNK(HAS_NEWER) wait_vmcnt_le((HAS_NEWER) ? NG_WAVE - 1 : 0)

    const int sl_start = int((blockIdx.x + (blockIdx.x >> 3)) % unsigned(n_slices));   // rotated slice order (see above)
#define FD_SL(I) (slice0 + ((I) + sl_start) % n_slices)
    constexpr int CHUNKS_PER_SLICE = 2 * NCHUNK;
    // chunk q of this workgroup's sequence: slice q / CHUNKS_PER_SLICE, pass (q / NCHUNK) & 1, column block q % NCHUNK
#define FD_DMA_SEQ(Q, BUFI) FD_DMA(FD_SL((Q) / CHUNKS_PER_SLICE), ((Q) / NCHUNK) & 1, (Q) % NCHUNK, BUFI)
    const int total_chunks = n_slices * CHUNKS_PER_SLICE;
    FD_DMA_SEQ(0, 0)                                      // start the weight stream before the activation slab
    if (total_chunks > 1) { FD_DMA_SEQ(1, 1) }

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
    if (RECUR) {
        for (int i = tid; i < BM; i += THREADS) {
            const int64_t b = int64_t(blockIdx.x) * BM + i;
            s_keep[i] = (keep && b < B) ? keep[b] : 1.0f;
        }
    }
    wait_vmcnt_le(0);                                     // chunks 0 and 1 landed (and the slab is in registers)
    __syncthreads();

    int q = 0;                                            // chunk counter; buffers rotate cur -> nx1 -> nx2
    int cur = 0, nx1 = 1, nx2 = 2;
    for (int si = 0; si < n_slices; ++si) {
        const int sl = FD_SL(si);
        const int col = sl * NSLICE + r;
        float ig[16];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            f32x16_t acc0, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ++ch, ++q) {
                if (q + 2 < total_chunks) {                   // chunk q + 2: requested two chunk-times before it is multiplied
                    const int lin2 = pass * NCHUNK + ch + 2;            // folds: pass and ch are unrolled
                    FD_DMA(FD_SL(si + lin2 / CHUNKS_PER_SLICE), (lin2 % CHUNKS_PER_SLICE) / NCHUNK, lin2 % NCHUNK, nx2)
                }
                const uint16_t* wb = s_w + cur * BUF;
#pragma unroll
                for (int ks = 0; ks < KC_STEPS; ++ks) {
                    const bf16x8_t af = a[ch * KC_STEPS + ks];
                    if (RECUR || pass == 0) {                 // rows 0..31 of the chunk: gate i (pass 0) / f (pass 1)
                        const uint4 b0 = *reinterpret_cast<const uint4*>(wb + r * ROW + ks * 16 + hf * 8);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, b0), acc0, 0, 0, 0);
                    }
                    const uint4 b1 = *reinterpret_cast<const uint4*>(wb + (NSLICE + r) * ROW + ks * 16 + hf * 8);   // g / o
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, b1), acc1, 0, 0, 0);
                }
                FD_WAIT_PREV_CHUNK(q + 2 < total_chunks);     // this wave's part of chunk q + 1 is in LDS
                __syncthreads();                              // ... and everyone's; buffer `cur` is free for chunk q + 3
                { const int t_ = cur; cur = nx1; nx1 = nx2; nx2 = t_; }
            }
            // C/D map of a 32x32 tile: col = lane & 31 (hidden unit), row = (e & 3) + 8 (e >> 2) + 4 hf
            if (pass == 0) {
                const float bi = bias[col] * -1.4426950408889634f, bg = bias[2 * H + col] * 2.8853900817779268f;
#pragma unroll
                for (int e = 0; e < 16; ++e) ig[e] = sigmoid_b(acc0[e], bi) * tanh_b(acc1[e], bg);
            } else {
                const float bo = bias[3 * H + col] * -1.4426950408889634f, bf = RECUR ? bias[H + col] * -1.4426950408889634f : 0.0f;
                // Addresses: [per-wave base pointer] + [one 32-bit per-lane offset] + [compile-time row constant * H].
                // Written as b * H + col per element, LLVM hoists sixteen 64-bit row addresses per output array out of
                // the slice loop and the activation slab spills.
                const bool full = row0 + 32 <= B;             // wave-uniform: every row of this wave's tile exists
                const int lane_off = hf * 4 * H + col;
                const int64_t wave_off = (row0 < B ? row0 : 0) * H;      // a wave wholly past B reads row 0, stores nothing
                const int rows_left = row0 < B ? int(B - row0 < 32 ? B - row0 : 32) : 0;   // rows of this tile that exist
                // Uniform base pointers + UNSIGNED 32-bit lane offsets: the loads / stores take the "scalar base + vector offset"
                // form (as signed ints every one of the 64 accesses paid a 64-bit shift-add pair), and the wave-uniform `full`
                // picks a branch-free store sequence for every tile except the batch's ragged last one.
                // (row0 recomputed from the readfirstlane'd wave index: the compiler then knows the bases are wave-uniform -> SGPRs)
                const int64_t urow0 = int64_t(blockIdx.x) * BM + uwave * 32;
                const int64_t uwave_off = (urow0 < B ? urow0 : 0) * H;
                const float* cp_base = RECUR ? c_prev + uwave_off : nullptr;
                float* c_base = c_out ? c_out + uwave_off : nullptr;
                float* h32_base = h_out_f32 ? h_out_f32 + uwave_off : nullptr;
                uint16_t* h_base = h_out + uwave_off;
                const uint32_t uoff = uint32_t(lane_off), uH = uint32_t(H), ucol = uint32_t(col);
                // BYTE offsets in 32 bits (a 32-row tile spans < 64 KB): with element indices the compiler must widen before scaling
                auto ld_f32 = [](const float* b, uint32_t boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + boff); };
                auto st_f32 = [](float* b, uint32_t boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(b) + boff) = v; };
                auto st_u16 = [](uint16_t* b, uint32_t boff, uint16_t v) { *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(b) + boff) = v; };
                float cp[16];                                 // all 16 c_prev loads issued together (clamped row)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
                    const uint32_t off = (full || lr < rows_left) ? uoff + uint32_t((e & 3) + 8 * (e >> 2)) * uH : ucol;
                    cp[e] = RECUR ? ld_f32(cp_base, off * 4u) : 0.0f;
                }
                float cv[16], hv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
                    float c = ig[e];
                    if (RECUR) c += sigmoid_b(acc0[e], bf) * (s_keep[wave * 32 + lr] * cp[e]);
                    cv[e] = c;
                    hv[e] = sigmoid_b(acc1[e], bo) * tanh_(c);
                }
                if (full) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t off = uoff + uint32_t((e & 3) + 8 * (e >> 2)) * uH;
                        if (c_base) st_f32(c_base, off * 4u, cv[e]);
                        st_u16(h_base, off * 2u, f2bf(hv[e]));
                        if (h32_base) st_f32(h32_base, off * 4u, hv[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int lr = (e & 3) + 8 * (e >> 2) + 4 * hf;
                        const uint32_t off = uoff + uint32_t((e & 3) + 8 * (e >> 2)) * uH;
                        if (lr < rows_left) {
                            if (c_base) st_f32(c_base, off * 4u, cv[e]);
                            st_u16(h_base, off * 2u, f2bf(hv[e]));
                            if (h32_base) st_f32(h32_base, off * 4u, hv[e]);
                        }
                    }
                }
            }
        }
    }
}

template <int KX, int KH>
int launch(const void* x, const void* h_prev, const float* c_prev, const float* keep, const void* W, const float* bias,
           void* h_out, float* c_out, float* h_out_f32, int64_t B, int H, hipStream_t st)
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
    hipLaunchKernelGGL((lstm_cell_mfma_dma_kernel<KX, KH>), dim3(unsigned(row_blocks), unsigned(split)), dim3(256), 0, st,
                       (const uint16_t*)x, (const uint16_t*)h_prev, c_prev, keep, (const uint16_t*)W, bias, (uint16_t*)h_out, c_out,
                       h_out_f32, B, H, split);
    return int(hipGetLastError());
}

}  // namespace

extern "C" int fdyn_lstm_cell_mfma(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                                   const void* W, const float* bias, void* h_out, float* c_out, float* h_out_f32,
                                   int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % NSLICE) return FDYN_ERR_BAD_SIZE;
    if (!x || !W || !bias || !h_out) return FDYN_ERR_NULL;
    if (kh > 0 && (!h_prev || !c_prev)) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (kx == 128 && kh == 256) return launch<128, 256>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 128 && kh == 0) return launch<128, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 256 && kh == 0) return launch<256, 0>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    if (kx == 128 && kh == 128) return launch<128, 128>(x, h_prev, c_prev, keep, W, bias, h_out, c_out, h_out_f32, B, H, st);
    return FDYN_ERR_BAD_SIZE;      // supported (input, recurrent) widths: (128,256) (128,0) (256,0) (128,128)
}
