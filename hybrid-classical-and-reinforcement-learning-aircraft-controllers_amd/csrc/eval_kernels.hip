// eval_kernels.hip -- per-episode rate-control evaluation metrics on the device (gfx950).
//
// Reference: learned_controllers/eval/metrics.py:95-362 (MetricsCalculator.compute_metrics and its helpers), fed by
// the recording loops of learned_controllers/eval_rate.py:70-118,170-233.  The reference walks one Python episode at a
// time; here one lane owns one episode of a recorded [T][..][N] trajectory block (word-major, so every load is a
// coalesced row read) and streams over it three times:
//   pass A  max |cmd| per axis, the steady-state command mean, RMSE / smoothness / return sums
//   pass B  settling run-length, overshoot, rise crossing, mean |error|      (need pass A's per-axis scalars)
//   pass C  mean |error| after the settling index                            (needs pass B's index)
// HBM-bound by construction: (6 S + 16 B + 1 S) per recorded step per pass; fp64 arithmetic throughout.
// Sums run sequentially (NumPy sums pairwise): means agree with the reference to ~1e-15 relative, times / flags exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fdyn.h"
#include "fdyn_core.hpp"

namespace {

using namespace fdyn;

template <typename S>
__global__ void __launch_bounds__(256)
rate_metrics_kernel(const double* __restrict__ times /*[T]*/, const S* __restrict__ rates /*[T][3][n]*/,
                    const S* __restrict__ commands /*[T][3][n]*/, const float* __restrict__ actions /*[T][n][4]*/,
                    const S* __restrict__ rewards /*[T][n]*/, const int32_t* __restrict__ lengths /*[n]*/,
                    double settling_threshold, int settle_steps, int T, int64_t n, double* __restrict__ out /*[FD_NM][n]*/)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int len = lengths[i];
    len = len < 0 ? 0 : (len > T ? T : len);
    if (len == 0) {
#pragma unroll
        for (int k = 0; k < FD_NM; ++k) out[int64_t(k) * n + i] = 0.0;
        return;
    }
    const double t_end = times[len - 1];
    const int n_ss = len / 5 > 1 ? len / 5 : 1;                           // metrics.py:274

    // ---- pass A -----------------------------------------------------------------------------------------------
    double maxc0 = 0.0, maxc1 = 0.0, maxc2 = 0.0, ss0 = 0.0, ss1 = 0.0, ss2 = 0.0;
    double sum_sq = 0.0, sum_diff = 0.0, sum_rew = 0.0;
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < len; ++t) {
        const int64_t row = int64_t(t) * 3 * n + i;
        const double c0 = double(commands[row]), c1 = double(commands[row + n]), c2 = double(commands[row + 2 * n]);
        const double r0 = double(rates[row]), r1 = double(rates[row + n]), r2 = double(rates[row + 2 * n]);
        maxc0 = fmax(maxc0, fabs(c0)); maxc1 = fmax(maxc1, fabs(c1)); maxc2 = fmax(maxc2, fabs(c2));
        const double e0 = c0 - r0, e1 = c1 - r1, e2 = c2 - r2;
        sum_sq += e0 * e0; sum_sq += e1 * e1; sum_sq += e2 * e2;          // :356-357, row-major order
        const float4 a = reinterpret_cast<const float4*>(actions)[int64_t(t) * n + i];
        if (t > 0) {                                                      // :334-338, surfaces only
            sum_diff += fabs(double(a.x) - double(prev.x));
            sum_diff += fabs(double(a.y) - double(prev.y));
            sum_diff += fabs(double(a.z) - double(prev.z));
        }
        prev = a;
        sum_rew += double(rewards[int64_t(t) * n + i]);
        if (t >= len - n_ss) { ss0 += c0; ss1 += c1; ss2 += c2; }
    }
    const int64_t mid = int64_t(len / 2) * 3 * n + i;                     // :236
    const double cm0 = double(commands[mid]), cm1 = double(commands[mid + n]), cm2 = double(commands[mid + 2 * n]);

    // ---- per-axis scalars -----------------------------------------------------------------------------------------
    struct Axis {
        double maxc, thr, sgn, cmd_ss, rise_thr, over, sum_abs, sum_tail;
        int run, settle_idx, rise_idx;
        bool skip, rise_on;
    };
    auto axis_init = [&](double maxc, double cmid, double ss_sum) {
        Axis A;
        A.maxc = maxc;
        A.skip = maxc < 0.01;                                             // :144-145
        const double rel = settling_threshold * maxc;
        A.thr = rel > 0.05 ? rel : 0.05;                                  // :201-203
        A.sgn = cmid > 0.0 ? 1.0 : (cmid < 0.0 ? -1.0 : 0.0);
        A.cmd_ss = ss_sum / double(n_ss);
        A.rise_on = !(fabs(A.cmd_ss) < 0.01);                             // :277-278
        A.rise_thr = 0.9 * A.cmd_ss;
        A.over = -1.0; A.sum_abs = 0.0; A.sum_tail = 0.0;
        A.run = 0; A.settle_idx = (settle_steps <= 0) ? 0 : -1; A.rise_idx = -1;
        return A;
    };
    Axis A0 = axis_init(maxc0, cm0, ss0), A1 = axis_init(maxc1, cm1, ss1), A2 = axis_init(maxc2, cm2, ss2);

    // ---- pass B -----------------------------------------------------------------------------------------------
    auto axis_step = [&](Axis& A, double c, double r, int t) {
        const double e = c - r;
        const double ae = fabs(e);
        A.run = ae < A.thr ? A.run + 1 : 0;                               // :206-214: window [t-ss+1, t] all inside
        if (A.settle_idx < 0 && A.run >= settle_steps && t < len - 1) A.settle_idx = t - settle_steps + 1;
        if (A.sgn * (r - c) > 0.0) A.over = fmax(A.over, ae);             // :243-248
        if (A.rise_idx < 0 && (A.cmd_ss > 0.0 ? r >= A.rise_thr : r <= A.rise_thr)) A.rise_idx = t;   // :284-292
        A.sum_abs += ae;
    };
    for (int t = 0; t < len; ++t) {
        const int64_t row = int64_t(t) * 3 * n + i;
        axis_step(A0, double(commands[row]), double(rates[row]), t);
        axis_step(A1, double(commands[row + n]), double(rates[row + n]), t);
        axis_step(A2, double(commands[row + 2 * n]), double(rates[row + 2 * n]), t);
    }

    // ---- pass C: mean |error| from the settling index on (:296-321) ----------------------------------------------
    auto tail_from = [&](const Axis& A) { return (A.settle_idx >= 0 && A.settle_idx < len - 1) ? A.settle_idx : len; };
    const int f0 = tail_from(A0), f1 = tail_from(A1), f2 = tail_from(A2);
    int first = f0 < f1 ? f0 : f1;
    first = first < f2 ? first : f2;
    for (int t = first; t < len; ++t) {
        const int64_t row = int64_t(t) * 3 * n + i;
        if (t >= f0) A0.sum_tail += fabs(double(commands[row]) - double(rates[row]));
        if (t >= f1) A1.sum_tail += fabs(double(commands[row + n]) - double(rates[row + n]));
        if (t >= f2) A2.sum_tail += fabs(double(commands[row + 2 * n]) - double(rates[row + 2 * n]));
    }

    auto axis_store = [&](const Axis& A, int from, int ax) {
        double settle = 0.0, over = 0.0, sserr = 0.0, rise = 0.0;
        if (!A.skip) {
            settle = A.settle_idx >= 0 ? times[A.settle_idx] : t_end;
            over = (A.sgn != 0.0 && A.over >= 0.0) ? (A.over / A.maxc) * 100.0 : 0.0;
            rise = A.rise_on ? (A.rise_idx >= 0 ? times[A.rise_idx] : t_end) : 0.0;
            sserr = from < len ? A.sum_tail / double(len - from) : A.sum_abs / double(len);
        }
        out[int64_t(FD_M_SETTLE_ROLL + ax) * n + i] = settle;
        out[int64_t(FD_M_OVERSHOOT_ROLL + ax) * n + i] = over;
        out[int64_t(FD_M_SSERR_ROLL + ax) * n + i] = sserr;
        out[int64_t(FD_M_RISE_ROLL + ax) * n + i] = rise;
        return settle;
    };
    const double s0 = axis_store(A0, f0, 0), s1 = axis_store(A1, f1, 1), s2 = axis_store(A2, f2, 2);
    out[int64_t(FD_M_SMOOTHNESS) * n + i] = sum_diff / double((len - 1) * 3);         // len == 1: 0/0 = NaN, as NumPy
    out[int64_t(FD_M_RMSE) * n + i] = sqrt(sum_sq / double(len * 3));
    out[int64_t(FD_M_SUCCESS) * n + i] = (s0 < t_end && s1 < t_end && s2 < t_end) ? 1.0 : 0.0;   // :171-176
    out[int64_t(FD_M_EPISODE_LENGTH) * n + i] = t_end;
    out[int64_t(FD_M_TOTAL_REWARD) * n + i] = sum_rew;
}

template <typename S>
int launch_metrics(const double* times, const S* rates, const S* commands, const float* actions, const S* rewards,
                   const int32_t* lengths, double settling_threshold, int settle_steps, int T, int64_t n, double* out,
                   void* stream)
{
    if (T < 0 || n < 0 || settle_steps < 0) return FDYN_ERR_BAD_SIZE;
    if (n == 0) return FDYN_OK;
    if (!lengths || !out || (T > 0 && (!times || !rates || !commands || !actions || !rewards))) return FDYN_ERR_NULL;
    const unsigned blocks = unsigned((n + 255) / 256);
    hipLaunchKernelGGL(rate_metrics_kernel<S>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, times, rates, commands,
                       actions, rewards, lengths, settling_threshold, settle_steps, T, n, out);
    return int(hipGetLastError());
}

// RateTrackingReward.compute + SettlingTimeBonus.compute over recorded sequences (rewards.py:75-137,193-221): the operation
// order of env_reward (fdyn_core.hpp), with the weights as parameters and the components written out.
template <typename S>
__global__ void __launch_bounds__(256)
rate_reward_seq_kernel(const S* __restrict__ errs /*[T][3][n]*/, const S* __restrict__ actions /*[T][4][n]*/,
                       const S* __restrict__ prev0 /*[4][n]*/, const S* __restrict__ flight /*[T][FD_NRF][n]*/,
                       const S* __restrict__ cmd /*[3][n]*/, const double* __restrict__ params /*[FD_NRW]*/,
                       S* __restrict__ rstate /*[FD_NRS][n]*/, S dt, int T, int64_t n, S* __restrict__ tracking,
                       S* __restrict__ components, S* __restrict__ settle, uint8_t* __restrict__ settled)
{
#pragma clang fp contract(off)
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const S w_track = S(params[FD_RW_TRACKING]), w_smooth = S(params[FD_RW_SMOOTHNESS]), w_stab = S(params[FD_RW_STABILITY]);
    const S w_osc = S(params[FD_RW_OSCILLATION]), w_surv = S(params[FD_RW_SURVIVAL]);
    const S thr = S(params[FD_RW_SETTLE_THRESHOLD]), min_t = S(params[FD_RW_MIN_SETTLE_TIME]), mult = S(params[FD_RW_BONUS_MULTIPLIER]);
    S perr[3], sc[3], c[3], prev[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        perr[k] = rstate[int64_t(FD_RS_PERR_P + k) * n + i];
        sc[k] = rstate[int64_t(FD_RS_SIGN_P + k) * n + i];
        c[k] = cmd[int64_t(k) * n + i];
        prev[k] = prev0[int64_t(k) * n + i];
    }
    S timer = rstate[int64_t(FD_RS_SETTLE_TIMER) * n + i], is_settled = rstate[int64_t(FD_RS_IS_SETTLED) * n + i];
    for (int t = 0; t < T; ++t) {
        S err[3], a[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            err[k] = errs[(int64_t(t) * 3 + k) * n + i];
            a[k] = actions[(int64_t(t) * 4 + k) * n + i];
        }
        const S airspeed = flight[(int64_t(t) * FD_NRF + FD_RF_AIRSPEED) * n + i];
        const S altitude = flight[(int64_t(t) * FD_NRF + FD_RF_ALTITUDE) * n + i];
        const S roll = flight[(int64_t(t) * FD_NRF + FD_RF_ROLL) * n + i], pitch = flight[(int64_t(t) * FD_NRF + FD_RF_PITCH) * n + i];
        const S tracking_error = (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]) / S(3);           // :76
        const S r_tracking = -w_track * tracking_error;
        const S d0 = a[0] - prev[0], d1 = a[1] - prev[1], d2 = a[2] - prev[2];
        const S r_smooth = -w_smooth * (d0 * d0 + (d1 * d1 + d2 * d2));                                  // np.sum of 3: a0 + (a1 + a2)
        const S roll_st = M<S>::exp(-M<S>::abs(roll) / deg2rad<S>(45.0));
        const S pitch_st = M<S>::exp(-M<S>::abs(pitch) / deg2rad<S>(30.0));
        const S as_st = clipv((airspeed - S(8)) / S(12), S(0), S(1));
        const S alt_st = clipv((altitude - S(10)) / S(90), S(0), S(1));
        const S r_stab = w_stab * ((roll_st + pitch_st + as_st + alt_st) / S(4));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool flip = (signv(err[k]) != signv(perr[k])) && (M<S>::abs(perr[k]) > S(0.01));
            sc[k] = S(0.9) * sc[k] + (flip ? S(1) : S(0));
            perr[k] = err[k];
            prev[k] = a[k];
        }
        const S r_osc = -w_osc * (sc[0] + (sc[1] + sc[2]));
        const S total = r_tracking + r_smooth + r_stab + r_osc + w_surv;
        bool now = true;                                                                                  // :193-209
#pragma unroll
        for (int k = 0; k < 3; ++k) now = now && (M<S>::abs(err[k]) < pymax(M<S>::abs(c[k]) * thr, S(0.05)));
        S bonus = S(0);
        if (now) {
            timer += dt;
            if (timer >= min_t) { is_settled = S(1); bonus = mult * dt; }
        } else {
            timer = S(0);
            is_settled = S(0);
        }
        if (tracking) tracking[int64_t(t) * n + i] = total;
        if (components) {
            components[(int64_t(t) * FD_NRC + FD_RC_TRACKING) * n + i] = r_tracking;
            components[(int64_t(t) * FD_NRC + FD_RC_SMOOTHNESS) * n + i] = r_smooth;
            components[(int64_t(t) * FD_NRC + FD_RC_STABILITY) * n + i] = r_stab;
            components[(int64_t(t) * FD_NRC + FD_RC_OSCILLATION) * n + i] = r_osc;
            components[(int64_t(t) * FD_NRC + FD_RC_SURVIVAL) * n + i] = w_surv;
        }
        if (settle) settle[int64_t(t) * n + i] = bonus;
        if (settled) settled[int64_t(t) * n + i] = is_settled != S(0) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rstate[int64_t(FD_RS_PERR_P + k) * n + i] = perr[k];
        rstate[int64_t(FD_RS_SIGN_P + k) * n + i] = sc[k];
    }
    rstate[int64_t(FD_RS_SETTLE_TIMER) * n + i] = timer;
    rstate[int64_t(FD_RS_IS_SETTLED) * n + i] = is_settled;
}

template <typename S>
int launch_reward_seq(const S* errs, const S* actions, const S* prev0, const S* flight, const S* cmd, const double* params,
                      S* rstate, S dt, int T, int64_t n, S* tracking, S* components, S* settle, uint8_t* settled, void* stream)
{
    if (n < 0 || T < 0) return FDYN_ERR_BAD_SIZE;
    if (n == 0) return FDYN_OK;
    if (!params || !rstate || !cmd || !prev0 || (T > 0 && (!errs || !actions || !flight))) return FDYN_ERR_NULL;
    hipLaunchKernelGGL(rate_reward_seq_kernel<S>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, errs, actions,
                       prev0, flight, cmd, params, rstate, dt, T, n, tracking, components, settle, settled);
    return int(hipGetLastError());
}

}  // namespace

extern "C" {

int fdyn_rate_metrics_f64(const double* times, const double* rates, const double* commands, const float* actions,
                          const double* rewards, const int32_t* lengths, double settling_threshold, int settle_steps,
                          int T, int64_t n, double* out, void* stream)
{
    return launch_metrics<double>(times, rates, commands, actions, rewards, lengths, settling_threshold, settle_steps, T, n,
                                  out, stream);
}

int fdyn_rate_metrics_f32(const double* times, const float* rates, const float* commands, const float* actions,
                          const float* rewards, const int32_t* lengths, double settling_threshold, int settle_steps,
                          int T, int64_t n, double* out, void* stream)
{
    return launch_metrics<float>(times, rates, commands, actions, rewards, lengths, settling_threshold, settle_steps, T, n,
                                 out, stream);
}

int fdyn_rate_reward_seq_f64(const double* errs, const double* actions, const double* prev0, const double* flight,
                             const double* cmd, const double* params, double* rstate, double dt, int T, int64_t n,
                             double* tracking, double* components, double* settle, uint8_t* settled, void* stream)
{
    return launch_reward_seq<double>(errs, actions, prev0, flight, cmd, params, rstate, dt, T, n, tracking, components, settle, settled, stream);
}

int fdyn_rate_reward_seq_f32(const float* errs, const float* actions, const float* prev0, const float* flight,
                             const float* cmd, const double* params, float* rstate, float dt, int T, int64_t n,
                             float* tracking, float* components, float* settle, uint8_t* settled, void* stream)
{
    return launch_reward_seq<float>(errs, actions, prev0, flight, cmd, params, rstate, dt, T, n, tracking, components, settle, settled, stream);
}

}  // extern "C"
