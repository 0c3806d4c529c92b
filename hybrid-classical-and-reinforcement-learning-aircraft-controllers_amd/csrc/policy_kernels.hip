// policy_kernels.hip -- fused LSTM-cell point-wise kernels for the PPO+LSTM rate-controller policy (gfx950).
//
// The recurrent cell's contraction ([B, in+H] x [in+H, 4H]) runs on MFMA through hipBLASLt; what is left -- bias is
// already in the GEMM epilogue -- is the gate non-linearity and the cell update.  Un-fused, PyTorch spends 80 % of a
// 65 536-env policy step in ~40 element-wise launches over the [B, 4H] gate tensor (rocprofv3, profiles/); these
// kernels do the whole update in one pass: one 16-byte load per gate per lane, c/h written once.
//
// Gate order is PyTorch's (i, f, g, o) along the 4H axis.  Reference architecture: learned_controllers/networks/
// lstm_policy.py:49-61 (nn.LSTM(128, 256, 2)) and sb3_contrib's actor/critic nn.LSTM(128, 256).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/fdyn.h"
#include "philox.hpp"

namespace {

constexpr int VEC = 8;        // hidden units per lane: 8 x bf16 = 16 B, 8 x f32 = 2 x 16 B

struct bf16x8 { uint4 v; };

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(uint32_t(b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{   // round-to-nearest-even; NaN stays NaN (plain cast semantics)
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<uint16_t*>(&h);
}

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ float scalar(float v) { return v; }
    static __device__ __forceinline__ float round(float v) { return v; }
    static __device__ __forceinline__ void load(const float* p, float (&o)[VEC])
    {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&o)[VEC])
    {
        reinterpret_cast<float4*>(p)[0] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
};
template <> struct Vec8<uint16_t> {   // bf16 storage
    static __device__ __forceinline__ float scalar(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }
    static __device__ __forceinline__ float round(float v) { return bf2f(f2bf(v)); }      // the value a store keeps
    static __device__ __forceinline__ void load(const uint16_t* p, float (&o)[VEC])
    {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = { a.x, a.y, a.z, a.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(w[k] << 16); o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const float (&o)[VEC])
    {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = uint32_t(f2bf(o[2 * k])) | (uint32_t(f2bf(o[2 * k + 1])) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x)
{   // 1 - 2/(1+e^{2x}); saturates cleanly for |x| large
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// h, c <- LSTM cell update from pre-activation gates [B, 4H]; c_prev == nullptr means zero state (i, g, o only)
template <typename GT>
__global__ void __launch_bounds__(256)
lstm_cell_fwd_kernel(const GT* __restrict__ gates, const float* __restrict__ c_prev, float* __restrict__ h_f32,
                     GT* __restrict__ h_lp, float* __restrict__ c_out, GT* __restrict__ act_out, int64_t total_vec, int H,
                     const float* __restrict__ keep = nullptr /*[B]: c_prev *= keep (episode start)*/,
                     GT* __restrict__ h_next = nullptr /*row stride next_stride: h * keep_next, the next step's input*/,
                     int64_t next_stride = 0, const float* __restrict__ keep_next = nullptr,
                     const GT* __restrict__ bias = nullptr /*[rows / group_rows][4H]: added to the gates (batched GEMM has no epilogue)*/,
                     int64_t group_rows = 1)
{
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= total_vec) return;
    const int hv = H / VEC;
    const int64_t row = t / hv;
    const int j = int(t - row * hv) * VEC;
    const GT* g0 = gates + row * 4 * H + j;
    float gi[VEC], gf[VEC], gg[VEC], go[VEC], cp[VEC], h[VEC], c[VEC];
    Vec8<GT>::load(g0, gi);
    Vec8<GT>::load(g0 + 2 * H, gg);
    Vec8<GT>::load(g0 + 3 * H, go);
    if (c_prev) { Vec8<GT>::load(g0 + H, gf); Vec8<float>::load(c_prev + row * H + j, cp); }
    if (bias) {
        const GT* b0 = bias + (row / group_rows) * 4 * H + j;
        float bv[VEC];
        Vec8<GT>::load(b0, bv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) gi[k] += bv[k];
        Vec8<GT>::load(b0 + 2 * H, bv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) gg[k] += bv[k];
        Vec8<GT>::load(b0 + 3 * H, bv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) go[k] += bv[k];
        if (c_prev) {
            Vec8<GT>::load(b0 + H, bv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) gf[k] += bv[k];
        }
    }
    if (c_prev && keep) {
        const float kp = keep[row];
#pragma unroll
        for (int k = 0; k < VEC; ++k) cp[k] *= kp;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        gi[k] = sigmoidf_(gi[k]); gg[k] = tanhf_(gg[k]); go[k] = sigmoidf_(go[k]);
        if (c_prev) { gf[k] = sigmoidf_(gf[k]); c[k] = gf[k] * cp[k] + gi[k] * gg[k]; }
        else { gf[k] = 0.0f; c[k] = gi[k] * gg[k]; }
        h[k] = go[k] * tanhf_(c[k]);
    }
    if (c_out) Vec8<float>::store(c_out + row * H + j, c);
    if (h_f32) Vec8<float>::store(h_f32 + row * H + j, h);
    if (h_lp) Vec8<GT>::store(h_lp + row * H + j, h);
    if (h_next) {
        const float kn = keep_next ? keep_next[row] : 1.0f;
        float hm[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) hm[k] = h[k] * kn;
        Vec8<GT>::store(h_next + row * next_stride + j, hm);
    }
    if (act_out) {
        GT* a0 = act_out + row * 4 * H + j;
        Vec8<GT>::store(a0, gi); Vec8<GT>::store(a0 + H, gf); Vec8<GT>::store(a0 + 2 * H, gg); Vec8<GT>::store(a0 + 3 * H, go);
    }
}

// gradient of the cell update w.r.t. the pre-activation gates and c_prev.
// A block owns `rows_per_block` consecutive rows and walks them in passes of 256 * VEC / H rows, every lane keeping its
// column: with bias_ws != nullptr the lanes also sum their dgates over the passes, the block folds the passes' row slots
// through LDS and writes ONE partial row [4H] to bias_ws[blockIdx.x] -- the bias gradient (column sums of dgates over all
// rows of a BPTT pass) then costs a reduction over a few thousand partial rows instead of a second pass over the [rows, 4H]
// tensor (fdyn_colsum: 5 % of a PPO iteration, rocprofv3).  No atomics: graph-replay safe and deterministic.
// bias_ws needs 256 % (H / VEC) == 0 (lanes keep their column); without it any H % VEC == 0 works.
template <typename GT>
__global__ void __launch_bounds__(256)
lstm_cell_bwd_kernel(const GT* act /*may alias dgates: every lane reads its 4 x 8 values before it writes them*/,
                     const float* __restrict__ c_prev, const float* __restrict__ c_new,
                     const GT* __restrict__ dh, const float* __restrict__ dc_next, GT* dgates,
                     float* __restrict__ dc_prev, int64_t total_vec, int H, int64_t vecs_per_block, float* __restrict__ bias_ws,
                     const float* __restrict__ keep = nullptr /*[B]: the forward used keep * c_prev*/,
                     const GT* __restrict__ dh2 = nullptr /*second dh addend, row stride dh2_stride, scaled by dh2_keep*/,
                     int64_t dh2_stride = 0, const float* __restrict__ dh2_keep = nullptr,
                     const GT* __restrict__ pre_bias = nullptr /*non-null: `act` holds the PRE-activations (the GEMM's output, which
                     the forward then never had to re-write as activations: 8 of its 28 bytes per hidden unit); the bias
                     [rows / pre_group_rows][4H] is added and the gate non-linearities are re-evaluated here*/,
                     int64_t pre_group_rows = 1)
{
    __shared__ float s_red[256 * VEC * 4];
    const int hv = H / VEC;
    float acc[4][VEC];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[g][k] = 0.0f;
    const int64_t v0 = int64_t(blockIdx.x) * vecs_per_block;
    for (int64_t vo = threadIdx.x; vo < vecs_per_block; vo += 256) {
        const int64_t t = v0 + vo;
        if (t >= total_vec) break;
        const int64_t row = t / hv;
        const int j = int(t - row * hv) * VEC;
        const GT* a0 = act + row * 4 * H + j;
        float ai[VEC], af[VEC], ag[VEC], ao[VEC], cp[VEC], cn[VEC], dhv[VEC], dcn[VEC];
        float di[VEC], df[VEC], dg[VEC], dov[VEC], dcp[VEC];
        Vec8<GT>::load(a0, ai); Vec8<GT>::load(a0 + H, af); Vec8<GT>::load(a0 + 2 * H, ag); Vec8<GT>::load(a0 + 3 * H, ao);
        if (pre_bias) {
            const GT* b0 = pre_bias + (row / pre_group_rows) * 4 * H + j;
            float bi[VEC], bf_[VEC], bg[VEC], bo[VEC];
            Vec8<GT>::load(b0, bi); Vec8<GT>::load(b0 + H, bf_); Vec8<GT>::load(b0 + 2 * H, bg); Vec8<GT>::load(b0 + 3 * H, bo);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                ai[k] = sigmoidf_(ai[k] + bi[k]); af[k] = sigmoidf_(af[k] + bf_[k]);
                ag[k] = tanhf_(ag[k] + bg[k]); ao[k] = sigmoidf_(ao[k] + bo[k]);
            }
        }
        if (c_new) {
            Vec8<float>::load(c_new + row * H + j, cn);
        } else {                                              // zero-state cell whose c was not kept: c = i * g from the saved gates
#pragma unroll
            for (int k = 0; k < VEC; ++k) cn[k] = ai[k] * ag[k];
        }
        Vec8<GT>::load(dh + row * H + j, dhv);
        if (c_prev) Vec8<float>::load(c_prev + row * H + j, cp);
        if (dc_next) Vec8<float>::load(dc_next + row * H + j, dcn);
        const float kp = keep ? keep[row] : 1.0f;
        if (dh2) {                                            // recurrent gradient from step t + 1, summed in fp32
            float d2[VEC];
            Vec8<GT>::load(dh2 + row * dh2_stride + j, d2);
            const float k2 = dh2_keep ? dh2_keep[row] : 1.0f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) dhv[k] += k2 * d2[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (c_prev) cp[k] *= kp;
            const float tc = tanhf_(cn[k]);
            const float dc = dhv[k] * ao[k] * (1.0f - tc * tc) + (dc_next ? dcn[k] : 0.0f);
            dov[k] = dhv[k] * tc * ao[k] * (1.0f - ao[k]);
            di[k] = dc * ag[k] * ai[k] * (1.0f - ai[k]);
            dg[k] = dc * ai[k] * (1.0f - ag[k] * ag[k]);
            df[k] = c_prev ? dc * cp[k] * af[k] * (1.0f - af[k]) : 0.0f;
            dcp[k] = dc * af[k] * kp;
        }
        GT* d0 = dgates + row * 4 * H + j;
        Vec8<GT>::store(d0, di); Vec8<GT>::store(d0 + H, df); Vec8<GT>::store(d0 + 2 * H, dg); Vec8<GT>::store(d0 + 3 * H, dov);
        if (dc_prev) Vec8<float>::store(dc_prev + row * H + j, dcp);
        if (bias_ws) {                                        // the sums are of the values the weight-gradient GEMM will read
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                acc[0][k] += Vec8<GT>::round(di[k]); acc[1][k] += Vec8<GT>::round(df[k]);
                acc[2][k] += Vec8<GT>::round(dg[k]); acc[3][k] += Vec8<GT>::round(dov[k]);
            }
        }
    }
    if (bias_ws) {                                            // [row slot][4H] -> one partial row per block
        const int rs = threadIdx.x / hv, j = (threadIdx.x % hv) * VEC, rpp = 256 / hv;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_red[rs * 4 * H + g * H + j + k] = acc[g][k];
        __syncthreads();
        for (int c = threadIdx.x; c < 4 * H; c += 256) {
            float t = 0.0f;
            for (int q = 0; q < rpp; ++q) t += s_red[q * 4 * H + c];
            bias_ws[int64_t(blockIdx.x) * 4 * H + c] = t;
        }
    }
}

// ---- the zero-state cell in the THREE-gate layout (i, g, o along the 3H axis): the features extractor's two LSTM layers.
// The reference calls self.lstm(embedded) on a length-1 sequence without carried state (learned_controllers/networks/
// lstm_policy.py:75-92): h = c = 0 going in, so the forget gate multiplies zero and its pre-activation, its saved activation,
// its (zero) gradient and its rows of W_ih never need to exist -- a quarter of the layer's GEMM work and of these kernels'
// traffic (the rollout's policy_fe64.hip drops it the same way).
template <typename GT>
__global__ void __launch_bounds__(256)
lstm_cell0_fwd_kernel(const GT* __restrict__ gates /*[B][3H]*/, GT* __restrict__ h_out /*[B][H]*/, GT* act_out /*[B][3H] or null; may alias gates*/,
                      int64_t total_vec, int H)
{
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= total_vec) return;
    const int hv = H / VEC;
    const int64_t row = t / hv;
    const int j = int(t - row * hv) * VEC;
    const GT* g0 = gates + row * 3 * H + j;
    float gi[VEC], gg[VEC], go[VEC], h[VEC];
    Vec8<GT>::load(g0, gi); Vec8<GT>::load(g0 + H, gg); Vec8<GT>::load(g0 + 2 * H, go);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        gi[k] = sigmoidf_(gi[k]); gg[k] = tanhf_(gg[k]); go[k] = sigmoidf_(go[k]);
        h[k] = go[k] * tanhf_(gi[k] * gg[k]);
    }
    Vec8<GT>::store(h_out + row * H + j, h);
    if (act_out) {
        GT* a0 = act_out + row * 3 * H + j;
        Vec8<GT>::store(a0, gi); Vec8<GT>::store(a0 + H, gg); Vec8<GT>::store(a0 + 2 * H, go);
    }
}

// its gradient: dgates [B][3H] (may alias act) from the saved activations and dh; c = i * g is rebuilt (from the ROUNDED
// saved gates, as the four-gate kernel does for a zero-state cell); bias partial sums as in lstm_cell_bwd_kernel.
template <typename GT>
__global__ void __launch_bounds__(256)
lstm_cell0_bwd_kernel(const GT* act, const GT* __restrict__ dh, GT* dgates, int64_t total_vec, int H, int64_t vecs_per_block,
                      float* __restrict__ bias_ws)
{
    __shared__ float s_red[256 * VEC * 3];
    const int hv = H / VEC;
    float acc[3][VEC];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[g][k] = 0.0f;
    const int64_t v0 = int64_t(blockIdx.x) * vecs_per_block;
    for (int64_t vo = threadIdx.x; vo < vecs_per_block; vo += 256) {
        const int64_t t = v0 + vo;
        if (t >= total_vec) break;
        const int64_t row = t / hv;
        const int j = int(t - row * hv) * VEC;
        const GT* a0 = act + row * 3 * H + j;
        float ai[VEC], ag[VEC], ao[VEC], dhv[VEC], di[VEC], dg[VEC], dov[VEC];
        Vec8<GT>::load(a0, ai); Vec8<GT>::load(a0 + H, ag); Vec8<GT>::load(a0 + 2 * H, ao);
        Vec8<GT>::load(dh + row * H + j, dhv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float tc = tanhf_(ai[k] * ag[k]);
            const float dc = dhv[k] * ao[k] * (1.0f - tc * tc);
            dov[k] = dhv[k] * tc * ao[k] * (1.0f - ao[k]);
            di[k] = dc * ag[k] * ai[k] * (1.0f - ai[k]);
            dg[k] = dc * ai[k] * (1.0f - ag[k] * ag[k]);
        }
        GT* d0 = dgates + row * 3 * H + j;
        Vec8<GT>::store(d0, di); Vec8<GT>::store(d0 + H, dg); Vec8<GT>::store(d0 + 2 * H, dov);
        if (bias_ws) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                acc[0][k] += Vec8<GT>::round(di[k]); acc[1][k] += Vec8<GT>::round(dg[k]); acc[2][k] += Vec8<GT>::round(dov[k]);
            }
        }
    }
    if (bias_ws) {
        const int rs = threadIdx.x / hv, j = (threadIdx.x % hv) * VEC, rpp = 256 / hv;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < VEC; ++k) s_red[rs * 3 * H + g * H + j + k] = acc[g][k];
        __syncthreads();
        for (int c = threadIdx.x; c < 3 * H; c += 256) {
            float t = 0.0f;
            for (int q = 0; q < rpp; ++q) t += s_red[q * 3 * H + c];
            bias_ws[int64_t(blockIdx.x) * 3 * H + c] = t;
        }
    }
}

// GAE(lambda) over a [T, N] rollout, one lane per env, scanning t = T-1 .. 0 (rewards/values/starts row-major [T][N])
__global__ void __launch_bounds__(256)
gae_kernel(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ starts,
           const float* __restrict__ last_val, const float* __restrict__ last_done, float gamma, float lam, int T, int64_t N,
           float* __restrict__ adv, float* __restrict__ ret)
{
    const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float next_v = last_val[n], nonterm = 1.0f - last_done[n], run = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const float v = val[int64_t(t) * N + n];
        const float delta = rew[int64_t(t) * N + n] + gamma * next_v * nonterm - v;
        run = delta + gamma * lam * nonterm * run;
        adv[int64_t(t) * N + n] = run;
        ret[int64_t(t) * N + n] = run + v;
        next_v = v;
        nonterm = 1.0f - starts[int64_t(t) * N + n];
    }
}

// Diagonal-Gaussian action sampling + log-probability for a [B][4] mean (the policy head of the rate controller):
// actions = mean + exp(log_std) * N(0,1), log_prob = sum_k -0.5 z_k^2 - log_std_k - 0.5 log(2 pi).  One lane per env, one
// 16-byte load and store; the normals come from Philox-4x32-10 keyed by (seed, env, *step) (Box-Muller), where the step counter
// is a word in DEVICE memory that the caller bumps on the stream -- so a captured graph draws fresh noise on every replay.
// Replaces a dozen tiny element-wise / reduction launches per policy step.
template <typename MT>
__global__ void __launch_bounds__(256)
gaussian_head_kernel(const MT* __restrict__ mean /*[B][4]*/, const float* __restrict__ log_std /*[4]*/, uint64_t seed,
                     const uint32_t* __restrict__ step_ptr, int deterministic, float* __restrict__ actions /*[B][4]*/, float* __restrict__ logp /*[B]*/,
                     int64_t B)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float m[4];
    if constexpr (sizeof(MT) == 2) {
        const uint2 v = reinterpret_cast<const uint2*>(mean)[i];
        m[0] = __uint_as_float(v.x << 16); m[1] = __uint_as_float(v.x & 0xffff0000u);
        m[2] = __uint_as_float(v.y << 16); m[3] = __uint_as_float(v.y & 0xffff0000u);
    } else {
        const float4 v = reinterpret_cast<const float4*>(mean)[i];
        m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
    }
    float z[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (!deterministic) {
        uint32_t r[4];
        philox4(seed, uint32_t(i), uint32_t(i >> 32), step_ptr ? *step_ptr : 0u, 0x51u, r);
        const float u0 = (float(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = (float(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
        z[0] = ra * __cosf(6.283185307f * u1); z[1] = ra * __sinf(6.283185307f * u1);
        z[2] = rb * __cosf(6.283185307f * u3); z[3] = rb * __sinf(6.283185307f * u3);
    }
    float a[4], lp = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ls = log_std[k];
        a[k] = m[k] + __expf(ls) * z[k];
        lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
    }
    reinterpret_cast<float4*>(actions)[i] = make_float4(a[0], a[1], a[2], a[3]);
    logp[i] = lp;
}

// The two output heads of the policy fused with the sampling: mean = pi_hidden[B][64] Wa^T + ba (action_net, 64 -> 4),
// value = vf_hidden[B][64] wv + bv (value_net, 64 -> 1), then the Gaussian head above.  One lane per env: 2 x 128 B of bf16
// in, 16 + 4 + 4 B out.  As separate hipBLASLt GEMMs these N = 4 / N = 1 products took 16 + 7 us at B = 65 536.
__global__ void __launch_bounds__(256)
policy_heads_kernel(const uint16_t* __restrict__ pi_hidden /*[B][64] bf16*/, const uint16_t* __restrict__ vf_hidden /*[B][64]*/,
                    const uint16_t* __restrict__ Wa /*[4][64] bf16*/, const uint16_t* __restrict__ ba /*[4]*/,
                    const uint16_t* __restrict__ wv /*[64]*/, const uint16_t* __restrict__ bv /*[1]*/,
                    const float* __restrict__ log_std, uint64_t seed, const uint32_t* __restrict__ step_ptr, int deterministic,
                    float* __restrict__ actions, float* __restrict__ logp, float* __restrict__ value, int64_t B)
{
    __shared__ float s_w[5 * 64 + 5];
    for (int t = threadIdx.x; t < 5 * 64 + 5; t += blockDim.x) {
        float v;
        if (t < 256) v = bf2f(Wa[t]); else if (t < 320) v = bf2f(wv[t - 256]); else if (t < 324) v = bf2f(ba[t - 320]); else v = bf2f(bv[0]);
        s_w[t] = v;
    }
    __syncthreads();
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float m[4] = { s_w[320], s_w[321], s_w[322], s_w[323] }, val = s_w[324];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float hp[VEC], hv[VEC];
        Vec8<uint16_t>::load(pi_hidden + i * 64 + c * 8, hp);
        Vec8<uint16_t>::load(vf_hidden + i * 64 + c * 8, hv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = c * 8 + j;
            m[0] += hp[j] * s_w[k]; m[1] += hp[j] * s_w[64 + k]; m[2] += hp[j] * s_w[128 + k]; m[3] += hp[j] * s_w[192 + k];
            val += hv[j] * s_w[256 + k];
        }
    }
    float z[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (!deterministic) {
        uint32_t r[4];
        philox4(seed, uint32_t(i), uint32_t(i >> 32), step_ptr ? *step_ptr : 0u, 0x51u, r);
        const float u0 = (float(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = (float(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
        z[0] = ra * __cosf(6.283185307f * u1); z[1] = ra * __sinf(6.283185307f * u1);
        z[2] = rb * __cosf(6.283185307f * u3); z[3] = rb * __sinf(6.283185307f * u3);
    }
    float a[4], lp = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ls = log_std[k];
        a[k] = m[k] + __expf(ls) * z[k];
        lp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
    }
    reinterpret_cast<float4*>(actions)[i] = make_float4(a[0], a[1], a[2], a[3]);
    logp[i] = lp;
    value[i] = val;
}


// ---- reductions of the PPO update ---------------------------------------------------------------------------------------
// Written here rather than left to the framework's generic reductions for two reasons: they are few, large and fixed in
// shape (column sums of [T*B][N] gradient blocks, and the scalar sums of the clipped-surrogate loss), and the update is
// replayed from a hipGraph, where every kernel must be self-contained (the accumulators are cleared by a kernel of the
// same launch sequence, not by a memset node).
__global__ void zero_f32_kernel(float* __restrict__ p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

// Column sums of x [M][N] in two stages without atomics (deterministic): stage 1 -- block b reduces its contiguous chunk of
// rows to partial[b][0..N); stage 2 -- one lane per column adds the partials.  Wide rows (N % 8 == 0, N <= 2048): a lane
// owns 8 adjacent columns (16-byte loads, four in flight), the block's 256 lanes cover 256 / (N/8) rows per pass and meet
// in LDS.  Narrow rows (N in {1, 2, 4, 8}): the chunk is read as a flat stream, lane t always sees column t % N.
constexpr int COLSUM_BLOCKS = 1024;
template <typename GT>
__global__ void __launch_bounds__(256)
colsum_wide_kernel(const GT* __restrict__ x, int64_t M, int N, int64_t chunk, float* __restrict__ partial)
{
    __shared__ float s[256 * VEC];
    const int groups = N / VEC;                              // column groups per row (<= 256)
    const int rows_per_pass = 256 / groups;
    const int g = threadIdx.x % groups, rl = threadIdx.x / groups;
    const int64_t r0 = int64_t(blockIdx.x) * chunk;
    const int64_t r1 = r0 + chunk < M ? r0 + chunk : M;
    float acc[VEC], v0[VEC], v1[VEC], v2[VEC], v3[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    if (rl < rows_per_pass) {
        int64_t r = r0 + rl;
        const int64_t step = rows_per_pass;
        for (; r + 3 * step < r1; r += 4 * step) {
            Vec8<GT>::load(x + r * N + g * VEC, v0);
            Vec8<GT>::load(x + (r + step) * N + g * VEC, v1);
            Vec8<GT>::load(x + (r + 2 * step) * N + g * VEC, v2);
            Vec8<GT>::load(x + (r + 3 * step) * N + g * VEC, v3);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += (v0[k] + v1[k]) + (v2[k] + v3[k]);
        }
        for (; r < r1; r += step) {
            Vec8<GT>::load(x + r * N + g * VEC, v0);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += v0[k];
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[(rl * groups + g) * VEC + k] = acc[k];     // [rl][column]
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
        float t = 0.0f;
        for (int q = 0; q < rows_per_pass; ++q) t += s[q * N + c];
        partial[int64_t(blockIdx.x) * N + c] = t;
    }
}

template <typename GT>
__global__ void __launch_bounds__(256)
colsum_narrow_kernel(const GT* __restrict__ x, int64_t M, int N /*1, 2, 4 or 8*/, int64_t chunk, float* __restrict__ partial)
{
    __shared__ float s[256];
    const int64_t e0 = int64_t(blockIdx.x) * chunk * N;
    const int64_t e1 = (int64_t(blockIdx.x) * chunk + chunk < M ? int64_t(blockIdx.x) * chunk + chunk : M) * N;
    float a0 = 0.0f, a1 = 0.0f;
    int64_t e = e0 + threadIdx.x;
    for (; e + 256 < e1; e += 512) { a0 += Vec8<GT>::scalar(x[e]); a1 += Vec8<GT>::scalar(x[e + 256]); }
    if (e < e1) a0 += Vec8<GT>::scalar(x[e]);
    s[threadIdx.x] = a0 + a1;                                // e0 is a multiple of N and 256 % N == 0: lane t holds column t % N
    __syncthreads();
    if (threadIdx.x < N) {
        float t = 0.0f;
        for (int q = threadIdx.x; q < 256; q += N) t += s[q];
        partial[int64_t(blockIdx.x) * N + threadIdx.x] = t;
    }
}

// any other N: one lane per column, rows of the chunk in sequence (correct, not fast)
template <typename GT>
__global__ void __launch_bounds__(256)
colsum_generic_kernel(const GT* __restrict__ x, int64_t M, int N, int64_t chunk, float* __restrict__ partial)
{
    const int64_t r0 = int64_t(blockIdx.x) * chunk;
    const int64_t r1 = r0 + chunk < M ? r0 + chunk : M;
    for (int c = threadIdx.x; c < N; c += 256) {
        float t = 0.0f;
        for (int64_t r = r0; r < r1; ++r) t += Vec8<GT>::scalar(x[r * N + c]);
        partial[int64_t(blockIdx.x) * N + c] = t;
    }
}

// stage 2: a block owns 32 adjacent columns; its 8 row-slices of 32 lanes split the partial rows and meet in LDS
__global__ void __launch_bounds__(256)
colsum_final_kernel(const float* __restrict__ partial, int nb, int N, float* __restrict__ out)
{
    __shared__ float s[8][33];
    const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float a0 = 0.0f, a1 = 0.0f;
    if (c < N) {
        int b = sl;
        for (; b + 8 < nb; b += 16) { a0 += partial[int64_t(b) * N + c]; a1 += partial[int64_t(b + 8) * N + c]; }
        if (b < nb) a0 += partial[int64_t(b) * N + c];
    }
    s[sl][cx] = a0 + a1;
    __syncthreads();
    if (sl == 0 && c < N) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s[q][cx];
        out[c] = t;
    }
}

// stage 1.5 for MANY partial rows (the bias partials of the fused backward kernels: thousands of rows): block (x, y) sums rows
// [64 y, 64 y + 64) of 32 adjacent columns -> mid [gridDim.y][N]; colsum_final_kernel finishes.  (The final kernel alone
// would read the whole table with N / 32 blocks.)
__global__ void __launch_bounds__(256)
colsum_mid_kernel(const float* __restrict__ partial, int64_t nb, int N, float* __restrict__ mid)
{
    __shared__ float s[8][33];
    const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int64_t r0 = int64_t(blockIdx.y) * 64, r1 = r0 + 64 < nb ? r0 + 64 : nb;
    float a = 0.0f;
    if (c < N)
        for (int64_t b = r0 + sl; b < r1; b += 8) a += partial[b * N + c];
    s[sl][cx] = a;
    __syncthreads();
    if (sl == 0 && c < N) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s[q][cx];
        mid[int64_t(blockIdx.y) * N + c] = t;
    }
}

// ws[0] += sum adv, ws[1] += sum adv^2
__global__ void __launch_bounds__(256)
adv_moments_kernel(const float* __restrict__ adv, int64_t M, float* __restrict__ ws)
{
    __shared__ float s0[256], s1[256];
    float a = 0.0f, b = 0.0f;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < M; i += int64_t(gridDim.x) * blockDim.x) {
        const float v = adv[i];
        a += v; b += v * v;
    }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(ws + 0, s0[0]); atomicAdd(ws + 1, s1[0]); }
}

// Clipped-surrogate PPO loss of one slice and its gradient w.r.t. the policy outputs, one lane per (t, env) sample.
// stats: [0] policy loss, [1] value loss, [2] approx KL, [3] clip fraction, [4] total loss (all means over M; the
// entropy term is added by the host from log_std), [5..8] d loss / d log_std (the log-prob part).
// Semantics: learned_controllers/train_rate.py drives SB3's PPO.train -- ratio = exp(logp - old_logp), advantages
// normalised with the unbiased std, -min(A r, A clip(r)), mse value loss, KL estimate mean((r - 1) - log r).
__global__ void __launch_bounds__(256)
ppo_loss_kernel(const float* __restrict__ mean /*[M][4]*/, const float* __restrict__ actions /*[M][4]*/,
                const float* __restrict__ log_std /*[4]*/, const float* __restrict__ values, const float* __restrict__ old_logp,
                const float* __restrict__ adv, const float* __restrict__ ret, const float* __restrict__ old_values,
                const float* __restrict__ ws /*[2] adv moments*/, int normalize_adv, float clip_range, float clip_range_vf,
                float vf_coef, int64_t M, float* __restrict__ dmean /*[M][4]*/, float* __restrict__ dvalues /*[M]*/,
                float* __restrict__ stats /*[9]*/)
{
    __shared__ float red[9][256];
    const float inv_m = 1.0f / float(M);
    float a_mean = 0.0f, a_scale = 1.0f;
    if (normalize_adv) {
        a_mean = ws[0] * inv_m;
        const float var = (ws[1] - float(M) * a_mean * a_mean) / float(M > 1 ? M - 1 : 1);      // torch.std: unbiased
        a_scale = 1.0f / (sqrtf(var > 0.0f ? var : 0.0f) + 1e-8f);
    }
    float ls[4], iv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ls[k] = log_std[k]; iv[k] = __expf(-2.0f * ls[k]); }
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0f;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < M; i += int64_t(gridDim.x) * blockDim.x) {
        const float4 mu = reinterpret_cast<const float4*>(mean)[i], ac = reinterpret_cast<const float4*>(actions)[i];
        const float d[4] = { ac.x - mu.x, ac.y - mu.y, ac.z - mu.z, ac.w - mu.w };
        float logp = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) logp += -0.5f * d[k] * d[k] * iv[k] - ls[k] - 0.9189385332046727f;
        const float lr = logp - old_logp[i];
        const float r = __expf(lr);
        const float A = (adv[i] - a_mean) * a_scale;
        const float rc = fminf(fmaxf(r, 1.0f - clip_range), 1.0f + clip_range);
        const float s1 = A * r, s2 = A * rc;
        const float g_logp = (s1 <= s2) ? -A * r * inv_m : 0.0f;          // d(-min(s1, s2)) / d logp
        float v = values[i];
        float dv_scale = 1.0f;
        if (clip_range_vf > 0.0f) {                                       // values = old + clip(values - old)
            const float ov = old_values[i], dvv = v - ov;
            dv_scale = (dvv >= -clip_range_vf && dvv <= clip_range_vf) ? 1.0f : 0.0f;
            v = ov + fminf(fmaxf(dvv, -clip_range_vf), clip_range_vf);
        }
        const float ev = v - ret[i];
        dvalues[i] = vf_coef * 2.0f * ev * inv_m * dv_scale;
        float4 gm;
        gm.x = g_logp * d[0] * iv[0]; gm.y = g_logp * d[1] * iv[1]; gm.z = g_logp * d[2] * iv[2]; gm.w = g_logp * d[3] * iv[3];
        reinterpret_cast<float4*>(dmean)[i] = gm;
        acc[0] += -fminf(s1, s2); acc[1] += ev * ev; acc[2] += (r - 1.0f) - lr; acc[3] += (fabsf(r - 1.0f) > clip_range) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[5 + k] += g_logp * (d[k] * d[k] * iv[k] - 1.0f);
    }
    acc[4] = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 9; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x < 9) {
        const int k = threadIdx.x;
        float v = red[k][0];
        if (k < 4) v *= inv_m;                                              // means; [5..8] already carry 1/M through g_logp
        if (k == 4) v = (red[0][0] + vf_coef * red[1][0]) * inv_m;          // policy + vf_coef * value part of the loss
        atomicAdd(stats + k, v);
    }
}

// The glue between an env step and the next policy step of a rollout, ONE launch instead of five framework ones (or, cast,
// copy, 1 - x, counter += 1: 26 us of a 330 us rollout step at 65 536 envs): episode_start = terminated | truncated as fp32,
// keep = 1 - episode_start (what the recurrent cells mask their state with), and the device-side step counter of the action
// noise (policy_heads / gaussian_head draw Philox blocks keyed by it) moves on by one.
__global__ void __launch_bounds__(256)
episode_flags_kernel(const uint8_t* __restrict__ terminated, const uint8_t* __restrict__ truncated, float* __restrict__ episode_start,
                     float* __restrict__ keep, int32_t* __restrict__ counter, int64_t n)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i == 0 && counter) *counter += 1;
    if (i >= n) return;
    const bool done = (terminated[i] | truncated[i]) != 0;
    if (episode_start) episode_start[i] = done ? 1.0f : 0.0f;
    if (keep) keep[i] = done ? 0.0f : 1.0f;
}

inline unsigned blocks(int64_t n) { return unsigned((n + 255) / 256); }

// grouped variant: rows come in segments of group_rows, segment j belonging to group j % n_groups (the [T][G][B] row order
// of a multi-cell LSTM sequence); every stage-1 block lies inside one segment.  out [n_groups][N].
__global__ void __launch_bounds__(256)
colsum_final_grouped_kernel(const float* __restrict__ partial, int nb, int N, int blocks_per_segment, int n_groups,
                            float* __restrict__ out)
{
    __shared__ float s[8][33];
    const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int g = blockIdx.y;
    float a0 = 0.0f;
    if (c < N) {
        const int stride = blocks_per_segment * n_groups;
        for (int base = g * blocks_per_segment; base < nb; base += stride) {
            const int e = base + blocks_per_segment < nb ? base + blocks_per_segment : nb;
            for (int b = base + sl; b < e; b += 8) a0 += partial[int64_t(b) * N + c];
        }
    }
    s[sl][cx] = a0;
    __syncthreads();
    if (sl == 0 && c < N) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s[q][cx];
        out[int64_t(g) * N + c] = t;
    }
}

static int64_t colsum_chunk(int64_t M, int64_t group_rows, int n_groups)
{
    if (n_groups <= 1) {
        int64_t nb = (M + 255) / 256;
        nb = nb < 1 ? 1 : (nb > COLSUM_BLOCKS ? COLSUM_BLOCKS : nb);
        return (M + nb - 1) / nb;
    }
    int64_t d = group_rows < 512 ? group_rows : 512;        // largest divisor of group_rows that is <= 512
    while (group_rows % d) --d;
    return d;
}

template <typename GT>
static void launch_colsum(const GT* x, int64_t M, int N, int64_t group_rows, int n_groups, float* out, float* ws, hipStream_t st)
{
    const int64_t chunk = colsum_chunk(M, group_rows, n_groups);
    const int nb = int((M + chunk - 1) / chunk);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (N % VEC == 0 && N / VEC <= 256 && aligned)
        hipLaunchKernelGGL((colsum_wide_kernel<GT>), dim3(nb), dim3(256), 0, st, x, M, N, chunk, ws);
    else if (N == 1 || N == 2 || N == 4 || N == 8)
        hipLaunchKernelGGL((colsum_narrow_kernel<GT>), dim3(nb), dim3(256), 0, st, x, M, N, chunk, ws);
    else
        hipLaunchKernelGGL((colsum_generic_kernel<GT>), dim3(nb), dim3(256), 0, st, x, M, N, chunk, ws);
    if (n_groups <= 1)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 31) / 32), dim3(256), 0, st, ws, nb, N, out);
    else
        hipLaunchKernelGGL(colsum_final_grouped_kernel, dim3((N + 31) / 32, n_groups), dim3(256), 0, st, ws, nb, N,
                           int(group_rows / chunk), n_groups, out);
}


}  // namespace

extern "C" {

int fdyn_lstm_cell_fwd(const void* gates, int gates_bf16, const float* c_prev, float* h_f32, void* h_lp, float* c_out,
                       void* act_out, int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % VEC) return FDYN_ERR_BAD_SIZE;
    if (!gates) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * (H / VEC);
    if (gates_bf16)
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<uint16_t>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)gates, c_prev, h_f32, (uint16_t*)h_lp, c_out, (uint16_t*)act_out, tv, H);
    else
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<float>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)gates, c_prev, h_f32, (float*)h_lp, c_out, (float*)act_out, tv, H);
    return int(hipGetLastError());
}

int fdyn_lstm_cell_bwd(const void* act, int bf16, const float* c_prev, const float* c_new, const void* dh,
                       const float* dc_next, void* dgates, float* dc_prev, int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % VEC) return FDYN_ERR_BAD_SIZE;
    if (!act || !dh || !dgates || (!c_new && c_prev)) return FDYN_ERR_NULL;     // c_new may be NULL only for a zero-state cell
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * (H / VEC);
    if (bf16)
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<uint16_t>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)act, c_prev, c_new, (const uint16_t*)dh, dc_next, (uint16_t*)dgates, dc_prev, tv, H, int64_t(256), (float*)nullptr);
    else
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<float>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)act, c_prev, c_new, (const float*)dh, dc_next, (float*)dgates, dc_prev, tv, H, int64_t(256), (float*)nullptr);
    return int(hipGetLastError());
}

int fdyn_lstm_seq_fwd(const void* gates, int gates_bf16, const float* c_prev, const float* keep, void* h_lp, float* c_out,
                      void* act_out, void* h_next, int64_t next_stride, const float* keep_next, const void* bias,
                      int64_t group_rows, int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % VEC || (h_next && next_stride < H) || (bias && group_rows < 1)) return FDYN_ERR_BAD_SIZE;
    if (!gates || !c_prev || !h_lp || !c_out) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * (H / VEC);
    if (gates_bf16)
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<uint16_t>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)gates, c_prev, (float*)nullptr, (uint16_t*)h_lp, c_out, (uint16_t*)act_out, tv, H, keep,
                           (uint16_t*)h_next, next_stride, keep_next, (const uint16_t*)bias, group_rows);
    else
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<float>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)gates, c_prev, (float*)nullptr, (float*)h_lp, c_out, (float*)act_out, tv, H, keep,
                           (float*)h_next, next_stride, keep_next, (const float*)bias, group_rows);
    return int(hipGetLastError());
}

static int seq_bwd_launch(const void* act, int bf16, const float* c_prev, const float* keep, const float* c_new, const void* dh,
                          const void* dh2, int64_t dh2_stride, const float* dh2_keep, const float* dc_next, void* dgates,
                          float* dc_prev, float* bias_ws, int64_t rows_per_block, int64_t B, int H, void* stream,
                          const void* pre_bias = nullptr, int64_t pre_group_rows = 1)
{
    if (B < 0 || H <= 0 || H % VEC || (dh2 && dh2_stride < H)) return FDYN_ERR_BAD_SIZE;
    if (!act || !c_prev || !c_new || !dh || !dgates || !dc_prev) return FDYN_ERR_NULL;
    const int hv = H / VEC;
    if (bias_ws && (256 % hv || rows_per_block < 1 || (rows_per_block * hv) % 256)) return FDYN_ERR_BAD_SIZE;
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * hv;
    const int64_t vpb = bias_ws ? rows_per_block * hv : 256;
    const unsigned nblk = unsigned((tv + vpb - 1) / vpb);
    if (bf16)
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<uint16_t>), dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)act, c_prev, c_new, (const uint16_t*)dh, dc_next, (uint16_t*)dgates, dc_prev, tv, H, vpb, bias_ws,
                           keep, (const uint16_t*)dh2, dh2_stride, dh2_keep, (const uint16_t*)pre_bias, pre_group_rows);
    else
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<float>), dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                           (const float*)act, c_prev, c_new, (const float*)dh, dc_next, (float*)dgates, dc_prev, tv, H, vpb, bias_ws,
                           keep, (const float*)dh2, dh2_stride, dh2_keep, (const float*)pre_bias, pre_group_rows);
    return int(hipGetLastError());
}

int fdyn_lstm_seq_bwd(const void* act, int bf16, const float* c_prev, const float* keep, const float* c_new, const void* dh,
                      const void* dh2, int64_t dh2_stride, const float* dh2_keep, const float* dc_next, void* dgates,
                      float* dc_prev, int64_t B, int H, void* stream)
{
    return seq_bwd_launch(act, bf16, c_prev, keep, c_new, dh, dh2, dh2_stride, dh2_keep, dc_next, dgates, dc_prev, nullptr, 0, B, H, stream);
}

int fdyn_lstm_seq_bwd_bsum(const void* act, int bf16, const float* c_prev, const float* keep, const float* c_new, const void* dh,
                           const void* dh2, int64_t dh2_stride, const float* dh2_keep, const float* dc_next, void* dgates,
                           float* dc_prev, float* bias_ws, int64_t rows_per_block, int64_t B, int H, void* stream)
{
    if (!bias_ws) return FDYN_ERR_NULL;
    return seq_bwd_launch(act, bf16, c_prev, keep, c_new, dh, dh2, dh2_stride, dh2_keep, dc_next, dgates, dc_prev, bias_ws, rows_per_block,
                          B, H, stream);
}

int fdyn_lstm_seq_bwd_pre(const void* gates, int bf16, const void* bias, int64_t group_rows, const float* c_prev, const float* keep,
                          const float* c_new, const void* dh, const void* dh2, int64_t dh2_stride, const float* dh2_keep,
                          const float* dc_next, void* dgates, float* dc_prev, float* bias_ws, int64_t rows_per_block, int64_t B, int H,
                          void* stream)
{
    if (!bias || group_rows < 1) return bias ? FDYN_ERR_BAD_SIZE : FDYN_ERR_NULL;
    return seq_bwd_launch(gates, bf16, c_prev, keep, c_new, dh, dh2, dh2_stride, dh2_keep, dc_next, dgates, dc_prev, bias_ws, rows_per_block,
                          B, H, stream, bias, group_rows);
}

int fdyn_lstm_cell0_fwd(const void* gates, int bf16, void* h_out, void* act_out, int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % VEC) return FDYN_ERR_BAD_SIZE;
    if (!gates || !h_out) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * (H / VEC);
    if (bf16)
        hipLaunchKernelGGL((lstm_cell0_fwd_kernel<uint16_t>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)gates, (uint16_t*)h_out, (uint16_t*)act_out, tv, H);
    else
        hipLaunchKernelGGL((lstm_cell0_fwd_kernel<float>), dim3(blocks(tv)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)gates, (float*)h_out, (float*)act_out, tv, H);
    return int(hipGetLastError());
}

int fdyn_lstm_cell0_bwd(const void* act, int bf16, const void* dh, void* dgates, float* bias_ws, int64_t rows_per_block,
                        int64_t B, int H, void* stream)
{
    if (B < 0 || H <= 0 || H % VEC) return FDYN_ERR_BAD_SIZE;
    if (!act || !dh || !dgates) return FDYN_ERR_NULL;
    const int hv = H / VEC;
    if (bias_ws && (256 % hv || rows_per_block < 1 || (rows_per_block * hv) % 256)) return FDYN_ERR_BAD_SIZE;
    if (B == 0) return FDYN_OK;
    const int64_t tv = B * hv;
    const int64_t vpb = bias_ws ? rows_per_block * hv : 256;
    const unsigned nblk = unsigned((tv + vpb - 1) / vpb);
    if (bf16)
        hipLaunchKernelGGL((lstm_cell0_bwd_kernel<uint16_t>), dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)act, (const uint16_t*)dh, (uint16_t*)dgates, tv, H, vpb, bias_ws);
    else
        hipLaunchKernelGGL((lstm_cell0_bwd_kernel<float>), dim3(nblk), dim3(256), 0, (hipStream_t)stream,
                           (const float*)act, (const float*)dh, (float*)dgates, tv, H, vpb, bias_ws);
    return int(hipGetLastError());
}

int fdyn_episode_flags(const uint8_t* terminated, const uint8_t* truncated, float* episode_start, float* keep, int32_t* counter,
                       int64_t n, void* stream)
{
    if (n < 0) return FDYN_ERR_BAD_SIZE;
    if (!terminated || !truncated) return FDYN_ERR_NULL;
    hipLaunchKernelGGL(episode_flags_kernel, dim3(blocks(n < 1 ? 1 : n)), dim3(256), 0, (hipStream_t)stream, terminated, truncated,
                       episode_start, keep, counter, n);
    return int(hipGetLastError());
}

int fdyn_colsum_partials(const float* partial, int64_t nb, int N, float* out, float* ws_mid, void* stream)
{
    if (nb < 1 || N <= 0 || nb > (int64_t(1) << 30)) return FDYN_ERR_BAD_SIZE;
    if (!partial || !out || !ws_mid) return FDYN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nmid = (nb + 63) / 64;
    hipLaunchKernelGGL(colsum_mid_kernel, dim3((N + 31) / 32, unsigned(nmid)), dim3(256), 0, st, partial, nb, N, ws_mid);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 31) / 32), dim3(256), 0, st, ws_mid, int(nmid), N, out);
    return int(hipGetLastError());
}

int64_t fdyn_colsum_ws_floats(int64_t M, int N, int64_t group_rows, int n_groups)
{
    const int64_t chunk = colsum_chunk(M < 1 ? 1 : M, group_rows < 1 ? 1 : group_rows, n_groups);
    return ((M + chunk - 1) / chunk + 1) * int64_t(N);
}

int fdyn_colsum(const void* x, int bf16, int64_t M, int N, int64_t group_rows, int n_groups, float* out, float* ws, void* stream)
{
    if (M < 0 || N <= 0 || n_groups < 1) return FDYN_ERR_BAD_SIZE;
    if (n_groups > 1 && (group_rows < 1 || M % (group_rows * n_groups))) return FDYN_ERR_BAD_SIZE;
    if (!out || !ws || (M > 0 && !x)) return FDYN_ERR_NULL;
    if (M == 0) {
        hipLaunchKernelGGL(zero_f32_kernel, dim3((N * n_groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, N * n_groups);
    } else if (bf16) {
        launch_colsum<uint16_t>((const uint16_t*)x, M, N, group_rows, n_groups, out, ws, (hipStream_t)stream);
    } else {
        launch_colsum<float>((const float*)x, M, N, group_rows, n_groups, out, ws, (hipStream_t)stream);
    }
    return int(hipGetLastError());
}

int fdyn_ppo_loss(const float* mean, const float* actions, const float* log_std, const float* values, const float* old_logp,
                  const float* adv, const float* ret, const float* old_values, int normalize_adv, float clip_range,
                  float clip_range_vf, float vf_coef, int64_t M, float* dmean, float* dvalues, float* stats /*[FDYN_PPO_NSTATS]*/,
                  float* ws /*[2]*/, void* stream)
{
    if (M <= 0) return FDYN_ERR_BAD_SIZE;
    if (!mean || !actions || !log_std || !values || !old_logp || !adv || !ret || !dmean || !dvalues || !stats || !ws ||
        (clip_range_vf > 0.0f && !old_values)) return FDYN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, st, stats, FDYN_PPO_NSTATS);
    hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, st, ws, 2);
    const unsigned nb = unsigned(M / 256 < 1 ? 1 : (M / 256 > 1024 ? 1024 : M / 256));
    if (normalize_adv) hipLaunchKernelGGL(adv_moments_kernel, dim3(nb), dim3(256), 0, st, adv, M, ws);
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(nb), dim3(256), 0, st, mean, actions, log_std, values, old_logp, adv, ret, old_values,
                       ws, normalize_adv, clip_range, clip_range_vf, vf_coef, M, dmean, dvalues, stats);
    return int(hipGetLastError());
}

int fdyn_gaussian_head(const void* mean, int mean_bf16, const float* log_std, uint64_t seed, const uint32_t* step, int deterministic,
                       float* actions, float* logp, int64_t B, void* stream)
{
    if (B < 0) return FDYN_ERR_BAD_SIZE;
    if (!mean || !log_std || !actions || !logp) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    if (mean_bf16)
        hipLaunchKernelGGL((gaussian_head_kernel<uint16_t>), dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)mean,
                           log_std, seed, step, deterministic, actions, logp, B);
    else
        hipLaunchKernelGGL((gaussian_head_kernel<float>), dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, (const float*)mean,
                           log_std, seed, step, deterministic, actions, logp, B);
    return int(hipGetLastError());
}

int fdyn_policy_heads(const void* pi_hidden, const void* vf_hidden, const void* Wa, const void* ba, const void* wv, const void* bv,
                      const float* log_std, uint64_t seed, const uint32_t* step, int deterministic, float* actions, float* logp,
                      float* value, int64_t B, void* stream)
{
    if (B < 0) return FDYN_ERR_BAD_SIZE;
    if (!pi_hidden || !vf_hidden || !Wa || !ba || !wv || !bv || !log_std || !actions || !logp || !value) return FDYN_ERR_NULL;
    if (B == 0) return FDYN_OK;
    hipLaunchKernelGGL(policy_heads_kernel, dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)pi_hidden,
                       (const uint16_t*)vf_hidden, (const uint16_t*)Wa, (const uint16_t*)ba, (const uint16_t*)wv, (const uint16_t*)bv,
                       log_std, seed, step, deterministic, actions, logp, value, B);
    return int(hipGetLastError());
}

int fdyn_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
             const float* last_dones, float gamma, float lam, int T, int64_t N, float* adv, float* ret, void* stream)
{
    if (T < 0 || N < 0) return FDYN_ERR_BAD_SIZE;
    if (T == 0 || N == 0) return FDYN_OK;
    hipLaunchKernelGGL(gae_kernel, dim3(blocks(N)), dim3(256), 0, (hipStream_t)stream, rewards, values, episode_starts,
                       last_values, last_dones, gamma, lam, T, N, adv, ret);
    return int(hipGetLastError());
}

}  // extern "C"
