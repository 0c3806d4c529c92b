// fdyn_kernels.hip -- gfx950 kernels + the C-ABI of include/fdyn.h.
//
// Mapping: one lane = one aircraft / env, 256-thread workgroups (4 wave64).  State is structure-of-arrays
// ([word][N]) so every load/store of a wave is one contiguous 256 B (fp32) / 512 B (fp64) segment.
// The aircraft "coefficient table" (one FD_NP-word block per aircraft TYPE; the reference has closed-form
// coefficients, not lookup tables) plus the PID / cascade / waypoint tables are staged in LDS once per
// workgroup; each lane then pulls its own type's block into registers, so heterogeneous fleets cost nothing
// extra in the inner loop.  All sub-steps of a launch run out of registers: HBM traffic per launch is one
// read + one write of the state.  Episode ends are compacted with a wave ballot + mbcnt prefix and ONE
// atomic per wave into a dense event list (terminal observation, return, length) -- the only inter-lane
// communication in the path; there is no inter-workgroup communication at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "fdyn_core.hpp"
#include "../../include/fdyn.h"

using namespace fdyn;

#define FD_BLOCK 256
#define FD_MAX_TYPES 8
#define FD_WAVE 64

// ---------------------------------------------------------------------------------------------------------
// LDS staging helpers
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void stage(T* dst, const T* __restrict__ src, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// Parameter blocks -> LDS: the FD_NP_USED caller words are copied; meanwhile a few threads per aircraft type fill the block's
// derived words straight from global memory, one word per lane (Params::derive_lane).  One barrier (the caller's).  Kernels
// issue their per-aircraft global loads BEFORE calling this, so the HBM round trip of the state overlaps the staging chain
// (global -> LDS -> barrier -> LDS -> registers) instead of following it.
template <bool FAST>
__device__ __forceinline__ void stage_params(double* s_params, const double* __restrict__ params, int n_types)
{
    for (int i = threadIdx.x; i < n_types * FD_NP_USED; i += blockDim.x) {
        const int t = i / FD_NP_USED, k = i - t * FD_NP_USED;
        s_params[t * FD_NP_STAGED + k] = params[t * FD_NP + k];
    }
    constexpr int NDL = Params<double>::FD_ND_LANES;
    if (int(threadIdx.x) < n_types * NDL) {
        const int t = threadIdx.x / NDL;
        Params<double>::derive_lane<FAST>(threadIdx.x - t * NDL, params + t * FD_NP, s_params + t * FD_NP_STAGED);
    }
}

// The same in two halves, for a kernel whose launch is one burst of per-aircraft loads (the env step at one wave per SIMD): loads
// return IN ORDER, so a parameter word requested after the state is not in LDS -- and the barrier not passed -- before the whole
// burst has landed.  `early` issues the block's loads FIRST; `finish` stores them once they are back, ahead of the state.
struct StagedParamWords { double v0, v1, d; };
__device__ __forceinline__ StagedParamWords stage_params_early(const double* __restrict__ params, int n_types)
{
    StagedParamWords w;                                           // FD_MAX_TYPES * FD_NP_USED <= 2 * FD_BLOCK: two words per thread
    const int i0 = threadIdx.x, i1 = FD_BLOCK + threadIdx.x;
    const int t0 = i0 / FD_NP_USED, t1 = i1 / FD_NP_USED;
    w.v0 = i0 < n_types * FD_NP_USED ? params[t0 * FD_NP + (i0 - t0 * FD_NP_USED)] : 0.0;
    w.v1 = i1 < n_types * FD_NP_USED ? params[t1 * FD_NP + (i1 - t1 * FD_NP_USED)] : 0.0;
    constexpr int NDL = Params<double>::FD_ND_LANES;
    const int t = threadIdx.x / NDL;
    w.d = int(threadIdx.x) < n_types * NDL ? params[t * FD_NP + Params<double>::derive_source(threadIdx.x - t * NDL)] : 1.0;
    return w;
}
template <bool FAST>
__device__ __forceinline__ void stage_params_finish(double* s_params, const StagedParamWords& w, int n_types)
{
    const int i0 = threadIdx.x, i1 = FD_BLOCK + threadIdx.x;
    const int t0 = i0 / FD_NP_USED, t1 = i1 / FD_NP_USED;
    if (i0 < n_types * FD_NP_USED) s_params[t0 * FD_NP_STAGED + (i0 - t0 * FD_NP_USED)] = w.v0;
    if (i1 < n_types * FD_NP_USED) s_params[t1 * FD_NP_STAGED + (i1 - t1 * FD_NP_USED)] = w.v1;
    constexpr int NDL = Params<double>::FD_ND_LANES;
    if (int(threadIdx.x) < n_types * NDL) {
        const int t = threadIdx.x / NDL;
        Params<double>::derive_lane_from<FAST>(threadIdx.x - t * NDL, w.d, s_params + t * FD_NP_STAGED);
    }
}

// Cascade constants -> LDS in the glue type, plus the two derived reciprocals the fp32 guidance uses
template <typename G>
__device__ __forceinline__ void stage_cascade_consts(G* s_consts, const double* __restrict__ consts)
{
    for (int k = threadIdx.x; k < FD_NC; k += blockDim.x) s_consts[k] = G(consts[k]);
    if (threadIdx.x < 2) {
        const double bank = consts[threadIdx.x == 0 ? FD_C_WP_MAX_BANK_RAD : FD_C_LOS_MAX_BANK_RAD];
        s_consts[FD_CD_WP_INV_G_TAN_BANK + threadIdx.x] = G(1.0 / (9.81 * ::tan(bank)));
    }
}

// Lane -> aircraft map: one lane = one aircraft, fully populated wave64.  (Round 1 measured half- and quarter-populated
// waves -- 32 / 16 lanes per wave to get 2 / 4 waves per SIMD at N = 65 536: 67.8 -> 99.9 / 182.7 us for the mixed env
// step.  The SIMD retires ~1 VALU per 5 cycles from one wave and ~1 per 3.4 from four, so doubling the instruction
// count to double the wave count loses; the plumbing was removed in round 2, the result stays in DESIGN.md.)
struct LaneMap { int64_t i, wave_first; int lane; bool on; };
__device__ __forceinline__ LaneMap lane_map(int64_t n)
{
    LaneMap m;
    m.lane = threadIdx.x & (FD_WAVE - 1);
    m.i = int64_t(blockIdx.x) * FD_BLOCK + threadIdx.x;
    m.wave_first = m.i - m.lane;
    m.on = m.i < n;
    return m;
}

__device__ __forceinline__ int lane_type(const uint8_t* __restrict__ type, int64_t i, int n_types)
{
    int t = type ? int(type[i]) : 0;
    return t < n_types ? t : n_types - 1;
}

// ---------------------------------------------------------------------------------------------------------
// K1: Simplified6DOF.step x n_sub                        (simplified_6dof.py:228-293, simulation_backend.py:95-99)
// ---------------------------------------------------------------------------------------------------------
template <typename S, typename T>
__global__ void __launch_bounds__(FD_BLOCK)
sixdof_step_kernel(S* __restrict__ xs, const S* __restrict__ us, const uint8_t* __restrict__ type,
                   const double* __restrict__ params, int n_types, int64_t n, S dt_sub, int n_sub,
                   S* __restrict__ derived_out)
{
    __shared__ double s_params[FD_MAX_TYPES * FD_NP_STAGED];
    const LaneMap lm = lane_map(n);
    const int64_t i = lm.i;
    S x[FD_NX], uc[FD_NU];
    int ty = 0;
    if (lm.on) {                                             // state / control loads in flight during the parameter staging
#pragma unroll
        for (int k = 0; k < FD_NX; ++k) x[k] = xs[k * n + i];
#pragma unroll
        for (int k = 0; k < FD_NU; ++k) uc[k] = us[k * n + i];
        ty = lane_type(type, i, n_types);
    }
    stage_params<sizeof(T) == 4>(s_params, params, n_types);
    __syncthreads();
    if (!lm.on) return;

    const double* blk = s_params + ty * FD_NP_STAGED;
    Params<T> P; P.load(blk);
    Limits<S> Lm; Lm.load(blk);
    Controls<T> C;
    C.set(P, uc[FD_U_ELEVATOR], uc[FD_U_AILERON], uc[FD_U_RUDDER], uc[FD_U_THROTTLE]);

    rk4_substeps<S, T>(P, Lm, C, x, dt_sub, n_sub);

#pragma unroll
    for (int k = 0; k < FD_NX; ++k) xs[k * n + i] = x[k];
    if (derived_out) {
        const Derived<S> d = derived<S>(x);
        derived_out[FD_D_AIRSPEED * n + i] = d.airspeed;
        derived_out[FD_D_ALTITUDE * n + i] = d.altitude;
        derived_out[FD_D_GROUND_SPEED * n + i] = d.ground_speed;
        derived_out[FD_D_HEADING * n + i] = d.heading;
    }
}

// get_state's derived scalars for a whole fleet (simplified_6dof.py:295-331)
template <typename S>
__global__ void __launch_bounds__(FD_BLOCK)
derived_kernel(const S* __restrict__ xs, int64_t n, S* __restrict__ out)
{
    const int64_t i = int64_t(blockIdx.x) * FD_BLOCK + threadIdx.x;
    if (i >= n) return;
    S x[FD_NX];
#pragma unroll
    for (int k = 0; k < FD_NX; ++k) x[k] = xs[k * n + i];
    const Derived<S> d = derived<S>(x);
    out[FD_D_AIRSPEED * n + i] = d.airspeed;
    out[FD_D_ALTITUDE * n + i] = d.altitude;
    out[FD_D_GROUND_SPEED * n + i] = d.ground_speed;
    out[FD_D_HEADING * n + i] = d.heading;
}

// ---------------------------------------------------------------------------------------------------------
// batched scalar PID (the reference's only native code, cpp/src/pid_controller.cpp:24-60), one lane per loop
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FD_BLOCK)
pid_batch_kernel(const float* __restrict__ cfg /*[8] shared or [n][8]*/, int cfg_per_lane,
                 float* __restrict__ state /*[3][n]*/, const float* __restrict__ setpoint,
                 const float* __restrict__ measurement, float dt, float* __restrict__ out, int64_t n)
{
    const int64_t i = int64_t(blockIdx.x) * FD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const PidCfg c = load_pid_cfg(cfg + (cfg_per_lane ? i * FD_NPC : 0), 0);
    PidState s{ state[FD_PS_INTEGRAL * n + i], state[FD_PS_ERR_PREV * n + i], state[FD_PS_DFILT * n + i] };
    out[i] = pid_compute(c, s, setpoint[i], measurement[i], dt);
    state[FD_PS_INTEGRAL * n + i] = s.integral;
    state[FD_PS_ERR_PREV * n + i] = s.err_prev;
    state[FD_PS_DFILT * n + i] = s.dfilt;
}

// ---------------------------------------------------------------------------------------------------------
// K1+K2: n_steps control steps of the 5-level cascade + one RK4 each
//        (examples/03_waypoint_square_demo.py:148-209 per aircraft; agents in fdyn_core.hpp)
// ---------------------------------------------------------------------------------------------------------
// Glue type of the agents: the storage type for the fp64 parity variant, the COMPUTE type for the fp32-evaluation variants
// (their PIDs take fp32 inputs anyway; round 1 ran the glue in fp64 -- ocml sincos / atan2 / fmod several times per control
// step -- and the glue cost more than the physics: 7.3 us per control step of which 3.1 us were the RK4).
// Storage type of the 21 env words (`e`): the state's storage type in the f64 and f32 variants, fp32 in `mixed` -- they are
// commands, previous actions and reward-tracker state, computed in fp32 there anyway (GlueOf below); as fp64 rows they were
// 272 of the 660 bytes an env step moves.  Two words are kept exact in fp32 form: the settling timer counts STEPS (fp32
// holds small integers exactly; the threshold is the step count at which the reference's accumulated fp64 sum reaches 0.2 s)
// and the episode time is step * dt rather than an accumulated fp32 sum.
template <typename S, typename T> struct EnvOf { using type = S; };
template <> struct EnvOf<double, float> { using type = float; };
template <typename S, typename T> struct GlueOf { using type = S; };
template <typename S> struct GlueOf<S, float> { using type = float; };

// (FD_BLOCK, 1): nine PID configurations, 27 PID states, the constants and the integrator state are ~250 live registers;
// under the default occupancy target the allocator parked part of them in AGPRs and paid ~80 v_accvgpr moves per control step
template <typename S, typename T>
__global__ void __launch_bounds__(FD_BLOCK, 1)
cascade_step_kernel(S* __restrict__ xs, float* __restrict__ pid_state /*[9*3][n]*/, int32_t* __restrict__ wp_idx,
                    const uint8_t* __restrict__ type, const double* __restrict__ params, int n_types,
                    const float* __restrict__ pid_cfg /*[9][8]*/, const double* __restrict__ consts /*[FD_NC]*/,
                    const double* __restrict__ wps /*[n_wp][4]*/, int n_wp, int64_t n, S dt, int n_steps,
                    S* __restrict__ surf_out /*[4][n] or null*/, int32_t* __restrict__ reached_total /*[n] or null*/)
{
    using G = typename GlueOf<S, T>::type;
    constexpr bool FAST = sizeof(T) == 4;
    __shared__ double s_params[FD_MAX_TYPES * FD_NP_STAGED];
    __shared__ float s_pid_cfg[FD_NPID * FD_NPC];
    __shared__ G s_consts[FD_NC_STAGED];
    __shared__ G s_wps[FD_MAX_WAYPOINTS * FD_NWP];
    const LaneMap lm = lane_map(n);
    const int64_t i = lm.i;
    PidState st[FD_NPID];
    S x[FD_NX];
    int32_t idx = 0;
    int ty = 0;
    if (lm.on) {                                             // per-aircraft loads in flight during the staging below
#pragma unroll
        for (int k = 0; k < FD_NPID; ++k)
            st[k] = PidState{ pid_state[(k * FD_NPS + FD_PS_INTEGRAL) * n + i], pid_state[(k * FD_NPS + FD_PS_ERR_PREV) * n + i],
                              pid_state[(k * FD_NPS + FD_PS_DFILT) * n + i] };
#pragma unroll
        for (int k = 0; k < FD_NX; ++k) x[k] = xs[k * n + i];
        idx = wp_idx[i];
        ty = lane_type(type, i, n_types);
    }
    stage_params<FAST>(s_params, params, n_types);
    stage(s_pid_cfg, pid_cfg, FD_NPID * FD_NPC);
    stage_cascade_consts<G>(s_consts, consts);
    for (int k = threadIdx.x; k < n_wp * FD_NWP; k += blockDim.x) s_wps[k] = G(wps[k]);
    __syncthreads();
    if (!lm.on) return;

    const double* blk = s_params + ty * FD_NP_STAGED;
    Params<T> P; P.load(blk);
    Limits<S> Lm; Lm.load(blk);
    PidCfg cfg[FD_NPID];
#pragma unroll
    for (int k = 0; k < FD_NPID; ++k) cfg[k] = load_pid_cfg(s_pid_cfg, k);
    G Cr[FD_NC_STAGED];                                      // cascade constants in registers: read once, not once per control step
#pragma unroll
    for (int k = 0; k < FD_NC_STAGED; ++k) Cr[k] = s_consts[k];
    int32_t reached = 0;
    const bool restart = int(Cr[FD_C_ON_COMPLETE]) == 1;
    Surfaces<G> surf{ G(0), G(0), G(0), G(0) };

    if constexpr (FAST) {
        FastRK f;
        f.init(x);
        const float hdt = float(S(0.5) * dt), fdt = float(dt), dt6 = float(dt / S(6));
        for (int s = 0; s < n_steps; ++s) {
            if (idx < n_wp && waypoint_reached<G>(Cr, s_wps + idx * FD_NWP, f.x0)) { idx += 1; reached += 1; }   // mission.update
            if (idx >= n_wp) { if (restart) idx = 0; else break; }          // COMPLETE: freeze this aircraft
            const Derived<G> d = derived_fast(f);
            surf = waypoint_agent<G>(cfg, st, Cr, s_wps + idx * FD_NWP, f.x0, d, fdt);
            Controls<T> C;
            C.set(P, surf.elevator, surf.aileron, surf.rudder, surf.throttle);
            // controlled flight: an aircraft pushing against the rate clamp or flying sideways is an exception here, not a
            // standing share of the fleet -- the special cases stay behind their wave-level branches (STRAIGHT = false)
            rk4_fast_step<S, false>(P, Lm, C, x, f, hdt, fdt, dt6);
        }
    } else {
        for (int s = 0; s < n_steps; ++s) {
            if (idx < n_wp && waypoint_reached<S>(Cr, s_wps + idx * FD_NWP, x)) { idx += 1; reached += 1; }   // mission.update
            if (idx >= n_wp) { if (restart) idx = 0; else break; }          // COMPLETE: freeze this aircraft
            const Derived<S> d = derived<S>(x);
            surf = waypoint_agent<S>(cfg, st, Cr, s_wps + idx * FD_NWP, x, d, dt);
            Controls<T> C;
            C.set(P, surf.elevator, surf.aileron, surf.rudder, surf.throttle);
            rk4_substeps<S, T>(P, Lm, C, x, dt, 1);
        }
    }

#pragma unroll
    for (int k = 0; k < FD_NX; ++k) xs[k * n + i] = x[k];
#pragma unroll
    for (int k = 0; k < FD_NPID; ++k) {
        pid_state[(k * FD_NPS + FD_PS_INTEGRAL) * n + i] = st[k].integral;
        pid_state[(k * FD_NPS + FD_PS_ERR_PREV) * n + i] = st[k].err_prev;
        pid_state[(k * FD_NPS + FD_PS_DFILT) * n + i] = st[k].dfilt;
    }
    wp_idx[i] = idx;
    if (reached_total) reached_total[i] += reached;
    if (surf_out) {
        surf_out[FD_U_ELEVATOR * n + i] = S(surf.elevator); surf_out[FD_U_AILERON * n + i] = S(surf.aileron);
        surf_out[FD_U_RUDDER * n + i] = S(surf.rudder); surf_out[FD_U_THROTTLE * n + i] = S(surf.throttle);
    }
}

// ---------------------------------------------------------------------------------------------------------
// One cascade LEVEL commanded directly, per aircraft: n_steps x {agent.compute_action(command, state, dt) -> set_controls ->
// one RK4 of dt}, or with n_steps == 0 just compute_action (no physics).  This is how the reference's tests and examples
// drive RateAgent / AttitudeAgent / HSAAgent / WaypointAgent (controllers/*_agent.py; closed-loop helper
// tests/test_control_integration.py:34-74).  cmd rows by level:
//   FD_LEVEL_RATE     p, q, r [rad/s], throttle            FD_LEVEL_HSA       heading [rad], speed [m/s], altitude [m], -
//   FD_LEVEL_ATTITUDE roll, pitch, yaw (NaN = none), thr   FD_LEVEL_WAYPOINT  north, east, altitude, speed (NaN = keep)
// ---------------------------------------------------------------------------------------------------------
template <typename S, typename T>
__global__ void __launch_bounds__(FD_BLOCK, 1)
agent_step_kernel(int level, S* __restrict__ xs, float* __restrict__ pid_state /*[9*3][n]*/, const uint8_t* __restrict__ type,
                  const double* __restrict__ params, int n_types, const float* __restrict__ pid_cfg,
                  const double* __restrict__ consts, const S* __restrict__ cmd /*[4][n]*/, int64_t n, S dt, int n_steps,
                  S* __restrict__ surf_out /*[4][n]*/, int cfg_per_lane /*pid_cfg is [n][9][8]: one gain set per aircraft*/)
{
    using G = typename GlueOf<S, T>::type;
    constexpr bool FAST = sizeof(T) == 4;
    __shared__ double s_params[FD_MAX_TYPES * FD_NP_STAGED];
    __shared__ float s_pid_cfg[FD_NPID * FD_NPC];
    __shared__ G s_consts[FD_NC_STAGED];
    stage_params<FAST>(s_params, params, n_types);
    if (!cfg_per_lane) stage(s_pid_cfg, pid_cfg, FD_NPID * FD_NPC);
    stage_cascade_consts<G>(s_consts, consts);
    __syncthreads();
    const LaneMap lm = lane_map(n);
    const int64_t i = lm.i;
    if (!lm.on) return;
    const double* blk = s_params + lane_type(type, i, n_types) * FD_NP_STAGED;
    Params<T> P; P.load(blk);
    Limits<S> Lm; Lm.load(blk);
    PidCfg cfg[FD_NPID];
    PidState st[FD_NPID];
#pragma unroll
    for (int k = 0; k < FD_NPID; ++k) {
        cfg[k] = cfg_per_lane ? load_pid_cfg(pid_cfg + i * (FD_NPID * FD_NPC), k) : load_pid_cfg(s_pid_cfg, k);
        st[k] = PidState{ pid_state[(k * FD_NPS + FD_PS_INTEGRAL) * n + i], pid_state[(k * FD_NPS + FD_PS_ERR_PREV) * n + i],
                          pid_state[(k * FD_NPS + FD_PS_DFILT) * n + i] };
    }
    S x[FD_NX];
#pragma unroll
    for (int k = 0; k < FD_NX; ++k) x[k] = xs[k * n + i];
    const G c0 = G(cmd[i]), c1 = G(cmd[n + i]), c2 = G(cmd[2 * n + i]), c3 = G(cmd[3 * n + i]);
    Surfaces<G> surf{ G(0), G(0), G(0), G(0) };
    const int iters = n_steps > 0 ? n_steps : 1;
    const G gdt = G(dt);
    // one agent call on the glue-typed state `xg` with its derived scalars `dd` (a callable so both precisions share the switch)
    auto act = [&](const G (&xg)[FD_NX], auto&& derive) {
        if (level == FD_LEVEL_RATE) {
            surf = rate_agent<G>(cfg, st, s_consts, c0, c1, c2, c3, xg, gdt);
        } else if (level == FD_LEVEL_ATTITUDE) {
            const bool has_yaw = c2 == c2;
            surf = attitude_agent<G>(cfg, st, s_consts, c0, c1, has_yaw ? c2 : G(0), has_yaw, c3, xg, gdt);
        } else if (level == FD_LEVEL_HSA) {
            surf = hsa_agent<G>(cfg, st, s_consts, c0, c1, c2, xg, derive(), gdt);
        } else {
            const G wp[FD_NWP] = { c0, c1, c2, c3 };
            surf = waypoint_agent<G>(cfg, st, s_consts, wp, xg, derive(), gdt);
        }
    };
    if constexpr (FAST) {
        FastRK f;
        f.init(x);
        const float hdt = float(S(0.5) * dt), fdt = float(dt), dt6 = float(dt / S(6));
        for (int s = 0; s < iters; ++s) {
            act(f.x0, [&]() { return derived_fast(f); });
            if (n_steps > 0) {
                Controls<T> C;
                C.set(P, surf.elevator, surf.aileron, surf.rudder, surf.throttle);
                rk4_fast_step<S, false>(P, Lm, C, x, f, hdt, fdt, dt6);
            }
        }
    } else {
        for (int s = 0; s < iters; ++s) {
            act(x, [&]() { return derived<S>(x); });
            if (n_steps > 0) {
                Controls<T> C;
                C.set(P, surf.elevator, surf.aileron, surf.rudder, surf.throttle);
                rk4_substeps<S, T>(P, Lm, C, x, dt, 1);
            }
        }
    }
    if (n_steps > 0) {
#pragma unroll
        for (int k = 0; k < FD_NX; ++k) xs[k * n + i] = x[k];
    }
#pragma unroll
    for (int k = 0; k < FD_NPID; ++k) {
        pid_state[(k * FD_NPS + FD_PS_INTEGRAL) * n + i] = st[k].integral;
        pid_state[(k * FD_NPS + FD_PS_ERR_PREV) * n + i] = st[k].err_prev;
        pid_state[(k * FD_NPS + FD_PS_DFILT) * n + i] = st[k].dfilt;
    }
    if (surf_out) {
        surf_out[FD_U_ELEVATOR * n + i] = S(surf.elevator); surf_out[FD_U_AILERON * n + i] = S(surf.aileron);
        surf_out[FD_U_RUDDER * n + i] = S(surf.rudder); surf_out[FD_U_THROTTLE * n + i] = S(surf.throttle);
    }
}

// ---------------------------------------------------------------------------------------------------------
// K3+K4: RateControlEnv.step for a whole vec-env, with done compaction and in-kernel auto-reset
// ---------------------------------------------------------------------------------------------------------
template <typename G> struct EnvConsts {
    G dt, mr0, mr1, mr2, scale;     // separate scalars: an indexed array here ends up in scratch
    int max_steps, cmd_type, n_sub;
    int settle_steps;               // consecutive settled steps at which the reference's accumulated timer reaches 0.2 s
    // arithmetic select: a ?: chain over struct fields is turned back into an indexed (scratch) load by the compiler
    __device__ __forceinline__ G max_rate(int ax) const { return mr0 * G(ax == 0) + mr1 * G(ax == 1) + mr2 * G(ax == 2); }
};

// `sched`: the four schedule words are live only for ramp / sine commands (rate_env.py:342-372); step and random-walk
// fleets neither read nor write them (64 B per env per step at fp64) -- they stay 0, which is what reset leaves there
template <typename G>
__device__ __forceinline__ void env_load(EnvState<G>& e, const G* __restrict__ es, int64_t n, int64_t i, bool sched = true)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e.cmd[k] = es[(FD_E_CMD_P + k) * n + i];
        e.prev_err[k] = es[(FD_E_PERR_P + k) * n + i];
        e.sign_changes[k] = es[(FD_E_SIGN_P + k) * n + i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e.prev_action[k] = es[(FD_E_PREV_AIL + k) * n + i];
        e.sched[k] = sched ? es[(FD_E_SCHED0 + k) * n + i] : G(0);
    }
    e.settle_timer = es[FD_E_SETTLE_TIMER * n + i]; e.is_settled = es[FD_E_IS_SETTLED * n + i];
    e.time = es[FD_E_TIME * n + i]; e.ep_return = es[FD_E_EP_RETURN * n + i];
}
template <typename G>
__device__ __forceinline__ void env_store(const EnvState<G>& e, G* __restrict__ es, int64_t n, int64_t i, bool sched = true)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        es[(FD_E_CMD_P + k) * n + i] = e.cmd[k];
        es[(FD_E_PERR_P + k) * n + i] = e.prev_err[k];
        es[(FD_E_SIGN_P + k) * n + i] = e.sign_changes[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        es[(FD_E_PREV_AIL + k) * n + i] = e.prev_action[k];
        if (sched) es[(FD_E_SCHED0 + k) * n + i] = e.sched[k];
    }
    es[FD_E_SETTLE_TIMER * n + i] = e.settle_timer; es[FD_E_IS_SETTLED * n + i] = e.is_settled;
    es[FD_E_TIME * n + i] = e.time; es[FD_E_EP_RETURN * n + i] = e.ep_return;
}

// one reset record drawn on the device (throughput mode): same distributions as
// learned_controllers/data/generators.py:47-72,74-125,190-214 and rate_env.py:306, Philox instead of MT19937
template <typename G>
__device__ __forceinline__ void device_reset_record(uint64_t seed, uint32_t env, uint32_t episode,
                                                    const EnvConsts<G>& ec, G (&rec)[FD_NR])
{
    Philox ph;
    uint32_t r0[4], r1[4], r2[4], r3[4];
    ph.block(seed, env, episode, 0u, 0u, r0);
    ph.block(seed, env, episode, 0u, 1u, r1);
    ph.block(seed, env, episode, 0u, 2u, r2);
    ph.block(seed, env, episode, 0u, 3u, r3);
    const float d15 = 0.26179938779914943f;     // radians(15)
    rec[FD_R_AIRSPEED] = G(15.0f + 15.0f * u01(r0[0]));
    rec[FD_R_ALTITUDE] = G(50.0f + 150.0f * u01(r0[1]));
    rec[FD_R_ROLL] = G(-d15 + 2.0f * d15 * u01(r0[2]));
    rec[FD_R_PITCH] = G(-d15 + 2.0f * d15 * u01(r0[3]));
    rec[FD_R_YAW] = G(6.283185307179586f * u01(r1[0]));
    rec[FD_R_P] = G(-0.1f + 0.2f * u01(r1[1]));
    rec[FD_R_Q] = G(-0.1f + 0.2f * u01(r1[2]));
    rec[FD_R_R] = G(-0.1f + 0.2f * u01(r1[3]));
    rec[FD_R_CMD0] = rec[FD_R_CMD1] = rec[FD_R_CMD2] = rec[FD_R_CMD3] = G(0);
    if (ec.cmd_type == FD_CMD_RANDOM_WALK) return;
    const bool sine = ec.cmd_type == FD_CMD_SINE;
    int k = sine ? 1 + int(u01(r2[0]) * 2.0f) : 1 + int(u01(r2[0]) * 3.0f);          // number of active axes
    k = k > 3 ? 3 : k;
    // Which axes are active: a random permutation (first, second, third) of the three axes, first k of it active.
    // Everything below is straight-line scalar selects with STATIC array indices: a run-time index into a private
    // array (or a ?: chain over array elements, which the compiler folds back into one) is mis-compiled / spilled.
    const int perm = int(u01(r2[1]) * 6.0f) % 6;
    const int first = perm >> 1;
    const int o1 = first == 2 ? 0 : first + 1, o2 = first == 0 ? 2 : first - 1;
    const int second = (perm & 1) ? o2 : o1;
    const float m0 = u01(r3[0]), m1 = u01(r3[1]), m2 = u01(r3[2]);
    const G mrs[3] = { ec.mr0, ec.mr1, ec.mr2 };
    G vals[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {                                  // a is a compile-time constant after unrolling
        const int pos = (a == first) ? 0 : ((a == second) ? 1 : 2);
        const float um = pos == 0 ? m0 : (pos == 1 ? m1 : m2);
        float mag = (0.3f + 0.7f * um) * float(ec.scale);
        if (sine) mag *= 0.5f;
        const float sgn = (!sine && ((r2[2] >> pos) & 1u)) ? -1.0f : 1.0f;
        vals[a] = (pos < k) ? G(sgn * mag) * mrs[a] : G(0);
    }
    rec[FD_R_CMD0] = vals[0]; rec[FD_R_CMD1] = vals[1]; rec[FD_R_CMD2] = vals[2];
    if (sine) rec[FD_R_CMD3] = G(0.1f + 1.9f * u01(r2[3]));
}

// RateControlEnv.reset body (rate_env.py:170-204) from one record
template <typename G, typename E>
__device__ __forceinline__ void env_apply_reset(const G (&rec)[FD_NR], int cmd_type, G (&x)[FD_NX], EnvState<E>& e)
{
#pragma unroll
    for (int k = 0; k < FD_NX; ++k) x[k] = G(0);
    x[2] = -rec[FD_R_ALTITUDE]; x[3] = rec[FD_R_AIRSPEED];
    x[6] = rec[FD_R_ROLL]; x[7] = rec[FD_R_PITCH]; x[8] = rec[FD_R_YAW];
    x[9] = rec[FD_R_P]; x[10] = rec[FD_R_Q]; x[11] = rec[FD_R_R];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e.cmd[k] = E(0); e.prev_err[k] = E(0); e.sign_changes[k] = E(0); e.prev_action[k] = E(0); }
#pragma unroll
    for (int k = 0; k < 4; ++k) e.sched[k] = E(0);
    e.prev_action[3] = E(0.5);                                                   // rate_env.py:193
    if (cmd_type == FD_CMD_STEP) {
        e.cmd[0] = E(rec[FD_R_CMD0]); e.cmd[1] = E(rec[FD_R_CMD1]); e.cmd[2] = E(rec[FD_R_CMD2]);
    } else if (cmd_type == FD_CMD_RAMP || cmd_type == FD_CMD_SINE) {
        e.sched[0] = E(rec[FD_R_CMD0]); e.sched[1] = E(rec[FD_R_CMD1]); e.sched[2] = E(rec[FD_R_CMD2]); e.sched[3] = E(rec[FD_R_CMD3]);
    }
    e.settle_timer = E(0); e.is_settled = E(0); e.time = E(0); e.ep_return = E(0);
}

template <typename G>
__device__ __forceinline__ void load_env_consts(const double* __restrict__ EC, EnvConsts<G>& ec)
{
    ec.dt = G(EC[FD_EC_DT]);
    ec.max_steps = int(EC[FD_EC_MAX_STEPS]);
    ec.cmd_type = int(EC[FD_EC_CMD_TYPE]);
    ec.scale = G(EC[FD_EC_DIFFICULTY_SCALE]);
    ec.mr0 = G(EC[FD_EC_MAX_RATE_P]); ec.mr1 = G(EC[FD_EC_MAX_RATE_Q]); ec.mr2 = G(EC[FD_EC_MAX_RATE_R]);
    const double r = EC[FD_EC_DT] / EC[FD_EC_DT_PHYSICS];                       // simulation_backend.py:95
    const long ns = long(r);
    ec.n_sub = ns < 1 ? 1 : int(ns);
    // rewards.py:205-209: `timer += dt; if timer >= 0.2` in fp64 -- 0.02 x 10 accumulates to 0.19999999999999998, so the bonus
    // starts at the 11th settled step.  The fp32 env words carry the step COUNT; this is the count the sum first reaches 0.2 at.
    const double dts = EC[FD_EC_DT];
    int k = 0;
    if (dts >= 1.0e-3) { double t = 0.0; while (t < 0.2 && k < 256) { t += dts; ++k; } }
    else k = int(0.2 / dts) + 1;
    ec.settle_steps = k;
}

template <typename G>
__device__ __forceinline__ void fetch_reset_record(const double* __restrict__ pool, int pool_depth, uint64_t seed,
                                                   int64_t i, int32_t episode, const EnvConsts<G>& ec, G (&rec)[FD_NR])
{
    if (pool) {
        const double* r = pool + (i * pool_depth + (episode % pool_depth)) * FD_NR;
#pragma unroll
        for (int k = 0; k < FD_NR; ++k) rec[k] = G(r[k]);
    } else {
        device_reset_record<G>(seed, uint32_t(i), uint32_t(episode), ec, rec);
    }
}

// coalesced write-out of a wave's 64 x 18 observation tile through LDS (row stride 19 words: conflict-free)
__device__ __forceinline__ void store_obs_tile(float* tile /*[64*19]*/, const float (&o)[FD_OBS_DIM], int lane,
                                               float* __restrict__ obs_out, int64_t wave_first, int64_t n)
{
    int64_t valid = n - wave_first;
    valid = valid < 0 ? 0 : (valid < FD_WAVE ? valid : FD_WAVE);
    float* dst = obs_out + wave_first * FD_OBS_DIM;
    static_assert(FD_OBS_DIM == 18 && FD_WAVE == 64, "the full-wave path below moves 64 x 18 floats as 288 float4");
    if (valid == FD_WAVE && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        // a full wave's tile is 4608 contiguous bytes of obs_out: rows go to LDS unpadded (nine 8-byte writes per lane) and
        // leave as 4.5 rounds of 16-byte loads / stores -- 9 + 5 + 5 memory instructions with no index arithmetic instead of
        // 18 + 18 + 18 with a division by 18 each (2.8 k of a wave's 96 k cycles per env step, scratch/phase_stamps.py)
        float2* row = reinterpret_cast<float2*>(tile + lane * FD_OBS_DIM);
#pragma unroll
        for (int k = 0; k < FD_OBS_DIM / 2; ++k) row[k] = make_float2(o[2 * k], o[2 * k + 1]);
        __builtin_amdgcn_wave_barrier();
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int j = 0; j < 4; ++j) d4[j * FD_WAVE + lane] = t4[j * FD_WAVE + lane];
        if (lane < FD_WAVE / 2) d4[4 * FD_WAVE + lane] = t4[4 * FD_WAVE + lane];
        __builtin_amdgcn_wave_barrier();
        return;
    }
#pragma unroll
    for (int k = 0; k < FD_OBS_DIM; ++k) tile[lane * (FD_OBS_DIM + 1) + k] = o[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < FD_OBS_DIM; ++k) {
        const int flat = k * FD_WAVE + lane;
        const int row = flat / FD_OBS_DIM, col = flat - row * FD_OBS_DIM;
        if (row < valid) dst[flat] = tile[row * (FD_OBS_DIM + 1) + col];
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename S, typename E>
__global__ void __launch_bounds__(FD_BLOCK)
rate_env_reset_kernel(S* __restrict__ xs, E* __restrict__ es, int32_t* __restrict__ eis, float* __restrict__ pid_state,
                      const uint8_t* __restrict__ mask, const double* __restrict__ EC, const double* __restrict__ pool,
                      int pool_depth, uint64_t seed, float* __restrict__ obs_out, int64_t n)
{
    __shared__ __attribute__((aligned(16))) float s_tile[FD_BLOCK / FD_WAVE][FD_WAVE * (FD_OBS_DIM + 1)];
    const LaneMap lm = lane_map(n);
    const int64_t i = lm.i;
    const int lane = lm.lane, wave = threadIdx.x / FD_WAVE;
    const bool active = lm.on;
    EnvConsts<S> ec;
    load_env_consts<S>(EC, ec);
    S x[FD_NX];
    EnvState<E> e;
    float o[FD_OBS_DIM];
#pragma unroll
    for (int k = 0; k < FD_OBS_DIM; ++k) o[k] = 0.0f;
    if (active) {
        const bool doit = mask ? mask[i] != 0 : true;
        if (doit) {
            const int32_t episode = eis[FD_EI_EPISODE * n + i];
            S rec[FD_NR];
            fetch_reset_record<S>(pool, pool_depth, seed, i, episode, ec, rec);
            env_apply_reset<S, E>(rec, ec.cmd_type, x, e);
#pragma unroll
            for (int k = 0; k < FD_NX; ++k) xs[k * n + i] = x[k];
            env_store<E>(e, es, n, i);
            eis[FD_EI_STEP * n + i] = 0;
            eis[FD_EI_EPISODE * n + i] = episode + 1;
            if (pid_state) for (int k = 0; k < 3 * FD_NPS; ++k) pid_state[k * n + i] = 0.0f;
        } else {
#pragma unroll
            for (int k = 0; k < FD_NX; ++k) x[k] = xs[k * n + i];
            env_load<E>(e, es, n, i);
        }
        S airspeed, altitude;
        airspeed_altitude<S>(x, airspeed, altitude);
        env_observation<S, S, E>(x, e, airspeed, altitude, o);
    }
    store_obs_tile(s_tile[wave], o, lane, obs_out, lm.wave_first, n);
}

#ifdef FD_PHASE_STAMPS
// timing experiment (scratch/phase_stamps.py; never in the shipped build): shader-clock stamps of wave 0 of every workgroup at
// five points of the env step -> where the 22 % of wave cycles that rocprofv3 reports as waiting are spent
__device__ unsigned long long fdyn_stamps[4096 * 8];
#define FD_STAMP(K) if (threadIdx.x == 0 && blockIdx.x < 4096) fdyn_stamps[blockIdx.x * 8 + (K)] = __builtin_readcyclecounter();
#define FD_STAMP_REAL(K) if (threadIdx.x == 0 && blockIdx.x < 4096) fdyn_stamps[blockIdx.x * 8 + (K)] = wall_clock64();   /* 100 MHz, chip-wide */
extern "C" int fdyn_debug_read_stamps(unsigned long long* out, int count)
{
    return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(fdyn_stamps), sizeof(unsigned long long) * size_t(count)));
}
extern "C" int fdyn_debug_read_counts(unsigned* out, int count, int clear)
{
    int rc = int(hipMemcpyFromSymbol(out, HIP_SYMBOL(fdyn_dbg_cnt), sizeof(unsigned) * size_t(count)));
    if (clear && rc == 0) {
        void* p = nullptr;
        rc = int(hipGetSymbolAddress(&p, HIP_SYMBOL(fdyn_dbg_cnt)));
        if (rc == 0) rc = int(hipMemset(p, 0, sizeof(unsigned) * 4096 * 8));
    }
    return rc;
}
#else
#define FD_STAMP(K)
#define FD_STAMP_REAL(K)
#endif

// OCC2: cap the registers at 256 so that two waves fit per SIMD.  At exactly one wave per SIMD (65 536 envs on 256 CUs) the
// uncapped allocation (258 VGPRs) is 4 % faster; past that the second wave hides the first one's issue gaps
// (1 Mi envs: 1.16e9 -> 1.56e9 env-steps/s).  The launcher picks by batch size; the arithmetic is the same.
template <typename S, typename T, bool OCC2>
__global__ void __launch_bounds__(FD_BLOCK, OCC2 ? 2 : 1)
rate_env_step_kernel(S* __restrict__ xs, typename EnvOf<S, T>::type* __restrict__ es, int32_t* __restrict__ eis,
                     const uint8_t* __restrict__ type, const double* __restrict__ params, int n_types,
                     const double* __restrict__ EC,
                     const float* __restrict__ actions /*[n][4] or null => in-kernel rate PID*/,
                     float* __restrict__ pid_state /*[3*3][n], PID mode*/, const float* __restrict__ pid_cfg /*[9][8]*/,
                     const double* __restrict__ casc_consts /*[FD_NC], PID mode*/, float* __restrict__ actions_out,
                     const S* __restrict__ rw_delta /*[3][n] or null*/,
                     const double* __restrict__ pool, int pool_depth, uint64_t seed, int auto_reset, float residual_scale,
                     float* __restrict__ obs_out /*[n][18]*/, float* __restrict__ reward_f32, S* __restrict__ reward_full,
                     uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                     int32_t* __restrict__ ev_count, int32_t* __restrict__ ev_count_next, int32_t* __restrict__ ev_int,
                     float* __restrict__ ev_flt, int ev_cap, int64_t n)
{
    __shared__ double s_params[FD_MAX_TYPES * FD_NP_STAGED];
    __shared__ __attribute__((aligned(16))) float s_tile[FD_BLOCK / FD_WAVE][FD_WAVE * (FD_OBS_DIM + 1)];
    __shared__ float s_pid_cfg[3 * FD_NPC];
    __shared__ S s_consts[FD_NC];
    FD_STAMP(0)
    FD_STAMP_REAL(7)
    const LaneMap lm = lane_map(n);
    const int64_t i = lm.i;
    const int lane = lm.lane, wave = threadIdx.x / FD_WAVE;
    const bool active = lm.on;
    EnvConsts<S> ec;
    load_env_consts<S>(EC, ec);
    const bool uses_sched = ec.cmd_type == FD_CMD_RAMP || ec.cmd_type == FD_CMD_SINE;
    // actions == null: the fused rate-PID demonstrator drives the env.  residual_scale > 0 (with actions AND pid state):
    // ResidualRateControlEnv -- action = clip(PID + scale * residual) (residual_rate_env.py:99-157).
    const bool residual_mode = actions != nullptr && residual_scale > 0.0f && pid_state != nullptr;
    const bool pid_mode = actions == nullptr || residual_mode;

    // ---- every per-env global load is issued HERE, before the parameter staging and its barrier: the HBM round trip of
    // the state overlaps the staging chain instead of following it
    using E = typename EnvOf<S, T>::type;
    S x[FD_NX];
    EnvState<E> e;
    int32_t step = 0, episode = 0;
    int ty = 0;
    float4 av = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // The reset record of an env that ends in this launch depends only on (seed, env, episode): with device sampling it is drawn
    // HERE, while the state loads are in flight -- at one wave per SIMD every wave of the launch loads at the same moment, the
    // 13 MB burst takes ~7 k cycles and the wave has nothing else to do, whereas at the end of the step the four Philox blocks
    // (80 quarter-rate 64-bit multiplies, ~2.5 k cycles) were paid by three waves in four, the slowest ones included.  The
    // episode counter is therefore loaded FIRST (loads return in order).  The register-capped build keeps the lazy draw: its
    // neighbour wave hides it, and 24 more live registers would spill.
    // Order of the requests = order of arrival: the parameter words (the barrier below waits for them), the episode counter (the
    // reset record), what the FIRST dynamics evaluation reads (velocities, angles, rates, the action), and only then the position
    // (first touched by the accumulate at the end of sub-step 1) and the env words (touched after the 20 sub-steps) -- the second
    // half of the burst lands under the first sub-step instead of in front of it.
    S rec_pre[FD_NR];
    const bool pre_drawn = !OCC2 && pool == nullptr && auto_reset != 0;
    // (fp32-evaluation builds only: in the fp64 build the split left a dead 20-byte private segment in the kernel descriptor --
    // never accessed, but a scratch set-up per launch -- and that build is the parity reference, not the benched one)
    constexpr bool EARLY_PARAMS = sizeof(T) == 4;
    StagedParamWords spw = { 0.0, 0.0, 1.0 };
    if constexpr (EARLY_PARAMS) spw = stage_params_early(params, n_types);
    if (active) {
        episode = eis[FD_EI_EPISODE * n + i];
#pragma unroll
        for (int k = 3; k < FD_NX; ++k) x[k] = xs[k * n + i];
        if (actions) av = reinterpret_cast<const float4*>(actions)[i];
        ty = lane_type(type, i, n_types);
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xs[k * n + i];
        step = eis[FD_EI_STEP * n + i];
        env_load<E>(e, es, n, i, uses_sched);
        if (pre_drawn) device_reset_record<S>(seed, uint32_t(i), uint32_t(episode), ec, rec_pre);
    }
    if constexpr (EARLY_PARAMS) stage_params_finish<true>(s_params, spw, n_types);
    else stage_params<false>(s_params, params, n_types);
    if (pid_mode) {
        stage(s_pid_cfg, pid_cfg, 3 * FD_NPC);
        for (int k = threadIdx.x; k < FD_NC; k += blockDim.x) s_consts[k] = S(casc_consts[k]);
    }
    __syncthreads();

    // double-buffered event counters: this launch appends to ev_count[] and clears the OTHER set for the next
    // launch, so the per-step host-side memset disappears from the stream
    if (ev_count_next && blockIdx.x == 0 && threadIdx.x < FD_EV_SHARDS) ev_count_next[threadIdx.x] = 0;

    float o[FD_OBS_DIM];
#pragma unroll
    for (int k = 0; k < FD_OBS_DIM; ++k) o[k] = 0.0f;
    bool done = false, term = false;
    S reward = S(0);

    if (active) {
        const double* blk = s_params + ty * FD_NP_STAGED;
        Params<T> P; P.load(blk);
        Limits<S> Lm; Lm.load(blk);

        // ---- action: from the policy ([n][4] f32, one 16-B load per lane) or the fused rate PID -------------
        float a_in[4];
        S res_bonus = S(0);
        if (!pid_mode) {
            a_in[0] = av.x; a_in[1] = av.y; a_in[2] = av.z; a_in[3] = av.w;
        } else {                                         // learned_controllers/utils/pid_demonstrations.py:47-77
            PidCfg cfg[3]; PidState st[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                cfg[k] = load_pid_cfg(s_pid_cfg, k);
                st[k] = PidState{ pid_state[(k * FD_NPS + 0) * n + i], pid_state[(k * FD_NPS + 1) * n + i], pid_state[(k * FD_NPS + 2) * n + i] };
            }
            const S pid_dt = s_consts[FD_C_PID_DT] > S(0) ? s_consts[FD_C_PID_DT] : ec.dt;
            const Surfaces<S> sf = rate_agent<S>(cfg, st, s_consts, S(e.cmd[0]), S(e.cmd[1]), S(e.cmd[2]), s_consts[FD_C_PID_THROTTLE], x, pid_dt);
            a_in[0] = float(sf.aileron); a_in[1] = float(sf.elevator); a_in[2] = float(sf.rudder); a_in[3] = float(sf.throttle);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pid_state[(k * FD_NPS + 0) * n + i] = st[k].integral; pid_state[(k * FD_NPS + 1) * n + i] = st[k].err_prev;
                pid_state[(k * FD_NPS + 2) * n + i] = st[k].dfilt;
            }
            if (residual_mode) {                      // float32 arithmetic, as the reference's NumPy float32 arrays
#pragma clang fp contract(off)
                const float rr[4] = { av.x, av.y, av.z, av.w };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float c = a_in[k] + rr[k] * residual_scale;
                    const float lo = k < 3 ? -1.0f : 0.0f;
                    a_in[k] = c < lo ? lo : (c > 1.0f ? 1.0f : c);
                }
                const float mag = rr[0] * rr[0] + (rr[1] * rr[1] + rr[2] * rr[2]);
                res_bonus = S(0.05) * (S(1) - S(mag) / S(3));   // small-correction bonus (:153-154); fp32 magnitude only
            }
        }
        if (actions_out) reinterpret_cast<float4*>(actions_out)[i] = make_float4(a_in[0], a_in[1], a_in[2], a_in[3]);
        S a[4];                                                                   // rate_env.py:225
        a[0] = clipv(S(a_in[0]), S(-1), S(1)); a[1] = clipv(S(a_in[1]), S(-1), S(1));
        a[2] = clipv(S(a_in[2]), S(-1), S(1)); a[3] = clipv(S(a_in[3]), S(0), S(1));

        // ---- sim.set_controls + sim.step(dt): n_sub RK4 sub-steps in registers (rate_env.py:228-237) --------
        Controls<T> C;
        C.set(P, a[1], a[0], a[2], a[3]);                                         // action = [ail, elev, rud, thr]
        const S dt_sub = ec.dt / S(ec.n_sub);
#ifdef FD_PHASE_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        FD_STAMP(1)
        rk4_substeps<S, T, !OCC2>(P, Lm, C, x, dt_sub, ec.n_sub);
        FD_STAMP(2)
        if constexpr (sizeof(E) == sizeof(S)) e.time += E(ec.dt);                 // :241-242 (the reference's accumulated sum)
        else e.time = E(S(step + 1) * ec.dt);                                     // fp32 env words: exact product, not an fp32 running sum
        step += 1;

        // ---- _update_command (rate_env.py:342-372) ------------------------------------------------------------
        if (ec.cmd_type == FD_CMD_RAMP) {
            const S t = S(e.time);
            if (t < S(3)) {
                const S alpha = t / S(3);
#pragma unroll
                for (int k = 0; k < 3; ++k) e.cmd[k] = E((S(1) - alpha) * S(0) + alpha * S(e.sched[k]));
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) e.cmd[k] = e.sched[k];
            }
        } else if (ec.cmd_type == FD_CMD_RANDOM_WALK) {
            S delta[3];
            if (rw_delta) {
#pragma unroll
                for (int k = 0; k < 3; ++k) delta[k] = rw_delta[k * n + i];
            } else {                                                              // generators.py:147-164 on device
                Philox ph; uint32_t r[4];
                ph.block(seed, uint32_t(i), uint32_t(episode), uint32_t(step), 7u, r);
                const float m0 = sqrtf(-2.0f * logf(u01(r[0]))), m1 = sqrtf(-2.0f * logf(u01(r[2])));
                const float n0 = m0 * cosf(6.283185307f * u01(r[1])), n1 = m0 * sinf(6.283185307f * u01(r[1]));
                const float n2 = m1 * cosf(6.283185307f * u01(r[3]));
                const S sd = S(0.1) * M<S>::sqrt(ec.dt) * ec.scale;
                delta[0] = S(n0) * sd * ec.mr0; delta[1] = S(n1) * sd * ec.mr1; delta[2] = S(n2) * sd * ec.mr2;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) e.cmd[k] = E(clipv(S(e.cmd[k]) + delta[k], -ec.max_rate(k), ec.max_rate(k)));
        } else if (ec.cmd_type == FD_CMD_SINE) {
            const S sn = M<S>::sin(S(2.0 * FD_PI) * S(e.sched[3]) * S(e.time));
#pragma unroll
            for (int k = 0; k < 3; ++k) e.cmd[k] = E(S(e.sched[k]) * sn);
        }

        // ---- reward, termination (rate_env.py:247-294,437-460) -------------------------------------------------
        // A = arithmetic type of the reward / observation glue: double in the fp64 parity variant, float where the
        // derivatives are evaluated in fp32 anyway (env_reward's header says why the settling timer is exempt)
        using A = typename GlueOf<S, T>::type;
        const A u_ = A(x[3]), v_ = A(x[4]), w_ = A(x[5]), roll_ = A(x[6]), pitch_ = A(x[7]);
        const A airspeed = M<A>::sqrt(u_ * u_ + v_ * v_ + w_ * w_), altitude = -A(x[2]);      // simplified_6dof.py:295-331
        const A err[3] = { A(e.cmd[0]) - A(x[9]), A(e.cmd[1]) - A(x[10]), A(e.cmd[2]) - A(x[11]) };
        const A aa[4] = { A(a[0]), A(a[1]), A(a[2]), A(a[3]) };
        A rew = env_reward<E, A>(e, err, aa, airspeed, altitude, roll_, pitch_, ec.dt, ec.settle_steps);
#pragma unroll
        for (int k = 0; k < 4; ++k) e.prev_action[k] = E(a[k]);                   // :282
        term = (altitude < A(5)) || (M<A>::abs(roll_) > deg2rad<A>(120.0)) || (M<A>::abs(pitch_) > deg2rad<A>(80.0)) ||
               (airspeed < A(8));
        const bool trunc = step >= ec.max_steps;
        if (term && !trunc) rew += A(-100);                                       // :289-292
        reward = S(rew) + res_bonus;
        e.ep_return += E(reward);
        done = term || trunc;
        env_observation<S, A, E>(x, e, airspeed, altitude, o);
        if (reward_f32) reward_f32[i] = float(reward);
        if (reward_full) reward_full[i] = reward;
        terminated[i] = term ? 1 : 0;
        truncated[i] = trunc ? 1 : 0;
    }

    FD_STAMP(3)
    // ---- K4: episode-done compaction -- wave ballot + mbcnt prefix, one atomic per wave ---------------------
    // The returning atomic is ISSUED here and its result consumed only after the auto-reset below: the reset work of the
    // finished lanes (Philox draws / pool reads, first observation) covers the atomic's round trip.
    const unsigned long long done_mask = __ballot(done);
    const bool compact = done_mask != 0ull && ev_count != nullptr;
    int base = 0, prefix = 0;
    const int shard = int(blockIdx.x) & (FD_EV_SHARDS - 1), cap_s = ev_cap / FD_EV_SHARDS;
    if (compact) {
        const int n_done = __popcll(done_mask);
        prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(done_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(done_mask), 0u));
        const int leader = __ffsll((long long)done_mask) - 1;
        if (lane == leader) base = atomicAdd(ev_count + shard, n_done);
    }
    float o_term[FD_OBS_DIM];                                                     // terminal observation of a finished episode
#pragma unroll
    for (int k = 0; k < FD_OBS_DIM; ++k) o_term[k] = o[k];
    const float ret_term = done ? float(e.ep_return) : 0.0f;
    const int32_t len_term = step;

    // ---- in-kernel auto-reset (vec-env semantics: the returned observation is the post-reset one) -----------
    if (active) {
        if (done && auto_reset) {
            S rec[FD_NR];
            if (pre_drawn) {
#pragma unroll
                for (int k = 0; k < FD_NR; ++k) rec[k] = rec_pre[k];
            } else {
                fetch_reset_record<S>(pool, pool_depth, seed, i, episode, ec, rec);
            }
            env_apply_reset<S, E>(rec, ec.cmd_type, x, e);
            step = 0;
            episode += 1;
            // first observation of the new episode, in the glue type of this variant (an fp64 square root here is ~400 cycles
            // that three waves in four pay once per launch in the steady state of a random-action fleet)
            using A0 = typename GlueOf<S, T>::type;
            A0 xa[FD_NX];
#pragma unroll
            for (int k = 0; k < FD_NX; ++k) xa[k] = A0(x[k]);
            A0 airspeed, altitude;
            airspeed_altitude<A0>(xa, airspeed, altitude);
            env_observation<S, A0, E>(x, e, airspeed, altitude, o);
            if (pid_mode) for (int k = 0; k < 3 * FD_NPS; ++k) pid_state[k * n + i] = 0.0f;   // pid_agent.reset()
        }
#pragma unroll
        for (int k = 0; k < FD_NX; ++k) xs[k * n + i] = x[k];
        env_store<E>(e, es, n, i, uses_sched);
        eis[FD_EI_STEP * n + i] = step;
        eis[FD_EI_EPISODE * n + i] = episode;
    }
    if (compact) {
        const int leader = __ffsll((long long)done_mask) - 1;
        base = __shfl(base, leader, FD_WAVE);
        const int local = base + prefix;
        if (done && local < cap_s) {
            const int slot = shard * cap_s + local;
            ev_int[slot * FD_EV_NI + FD_EV_ENV] = int32_t(i);
            ev_int[slot * FD_EV_NI + FD_EV_LENGTH] = len_term;
            ev_int[slot * FD_EV_NI + FD_EV_TERMINATED] = term ? 1 : 0;
            ev_flt[slot * FD_EV_NF] = ret_term;
#pragma unroll
            for (int k = 0; k < FD_OBS_DIM; ++k) ev_flt[slot * FD_EV_NF + 1 + k] = o_term[k];
        }
    }
    FD_STAMP(4)
    store_obs_tile(s_tile[wave], o, lane, obs_out, lm.wave_first, n);
    FD_STAMP(5)
#ifdef FD_PHASE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FD_STAMP_REAL(6)
}

// =========================================================================================================
// C-ABI (include/fdyn.h)
// =========================================================================================================
static inline unsigned grid_for(int64_t n) { return unsigned((n + FD_BLOCK - 1) / FD_BLOCK); }

static int g_simds = 0;
static int simd_count()
{
    if (!g_simds) {
        int dev = 0; hipDeviceProp_t p;
        g_simds = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount * 4 : 1024;
    }
    return g_simds;
}
static inline int launch_status() { return int(hipGetLastError()); }

#define FD_CHECK_COMMON(n, n_types)                                       \
    if ((n) < 0 || (n) > (int64_t(1) << 31) - FD_BLOCK) return FDYN_ERR_BAD_SIZE;   \
    if ((n_types) < 1 || (n_types) > FD_MAX_TYPES) return FDYN_ERR_BAD_TYPES;       \
    if ((n) == 0) return FDYN_OK;

extern "C" {

int fdyn_abi_version(void) { return FDYN_ABI_VERSION; }

int fdyn_num_substeps(double dt, double dt_physics)
{   // simulation/simulation_backend.py:95  max(1, int(dt / dt_physics))
    const double r = dt / dt_physics;
    const long n = long(r);
    return n < 1 ? 1 : int(n);
}

int64_t fdyn_event_capacity(int64_t n)
{   // every shard must hold all the envs of the workgroups that map to it
    const int64_t blocks = (n + FD_BLOCK - 1) / FD_BLOCK;
    return FD_EV_SHARDS * ((blocks + FD_EV_SHARDS - 1) / FD_EV_SHARDS) * FD_BLOCK;
}

int fdyn_device_info(int* cu_count, int* wave_size, char* arch, int arch_len)
{
    hipDeviceProp_t p;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return int(e);
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return int(e);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return FDYN_OK;
}

#define FD_DEFINE_SIXDOF(NAME, S, T)                                                                         \
    int NAME(S* x, const S* u, const uint8_t* type, const double* params, int n_types, int64_t n, double dt, \
             int n_sub, S* derived_out, void* stream)                                                        \
    {                                                                                                        \
        FD_CHECK_COMMON(n, n_types)                                                                          \
        if (n_sub < 1) return FDYN_ERR_BAD_SIZE;                                                             \
        const double dt_sub = dt / n_sub;                                                                    \
        /* simplified_6dof.py:241-245: dt <= min_timestep or > max_timestep raises ValueError (defaults) */ \
        if (!(dt_sub > 1e-6) || dt_sub > 1.0) return FDYN_ERR_BAD_DT;                                        \
        hipLaunchKernelGGL((sixdof_step_kernel<S, T>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           x, u, type, params, n_types, n, S(dt_sub), n_sub, derived_out);              \
        return launch_status();                                                                              \
    }
FD_DEFINE_SIXDOF(fdyn_sixdof_step_f64, double, double)
FD_DEFINE_SIXDOF(fdyn_sixdof_step_mixed, double, float)
FD_DEFINE_SIXDOF(fdyn_sixdof_step_f32, float, float)

int fdyn_derived_f64(const double* x, int64_t n, double* out, void* stream)
{
    FD_CHECK_COMMON(n, 1)
    hipLaunchKernelGGL((derived_kernel<double>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, x, n, out);
    return launch_status();
}
int fdyn_derived_f32(const float* x, int64_t n, float* out, void* stream)
{
    FD_CHECK_COMMON(n, 1)
    hipLaunchKernelGGL((derived_kernel<float>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, x, n, out);
    return launch_status();
}

int fdyn_pid_compute_batch(const float* cfg, int cfg_per_lane, float* state, const float* setpoint,
                           const float* measurement, float dt, float* out, int64_t n, void* stream)
{
    FD_CHECK_COMMON(n, 1)
    hipLaunchKernelGGL(pid_batch_kernel, dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, cfg, cfg_per_lane,
                       state, setpoint, measurement, dt, out, n);
    return launch_status();
}

#define FD_DEFINE_CASCADE(NAME, S, T)                                                                        \
    int NAME(S* x, float* pid_state, int32_t* wp_idx, const uint8_t* type, const double* params, int n_types, \
             const float* pid_cfg, const double* consts, const double* wps, int n_wp, int64_t n, double dt,  \
             int n_steps, S* surf_out, int32_t* reached_total, void* stream)                                 \
    {                                                                                                        \
        FD_CHECK_COMMON(n, n_types)                                                                          \
        if (n_wp < 1 || n_wp > FD_MAX_WAYPOINTS || n_steps < 0) return FDYN_ERR_BAD_SIZE;                    \
        if (!(dt > 1e-6) || dt > 1.0) return FDYN_ERR_BAD_DT;                                                \
        hipLaunchKernelGGL((cascade_step_kernel<S, T>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           x, pid_state, wp_idx, type, params, n_types, pid_cfg, consts, wps, n_wp, n, S(dt), \
                           n_steps, surf_out, reached_total);                                           \
        return launch_status();                                                                              \
    }
#define FD_DEFINE_AGENT(NAME, S, T)                                                                          \
    int NAME(int level, S* x, float* pid_state, const uint8_t* type, const double* params, int n_types,      \
             const float* pid_cfg, int cfg_per_lane, const double* consts, const S* cmd, int64_t n, double dt, \
             int n_steps, S* surf_out, void* stream)                                                         \
    {                                                                                                        \
        FD_CHECK_COMMON(n, n_types)                                                                          \
        if (level < FD_LEVEL_WAYPOINT || level > FD_LEVEL_RATE || n_steps < 0) return FDYN_ERR_BAD_SIZE;     \
        if (n > 0 && (!x || !pid_state || !pid_cfg || !consts || !cmd)) return FDYN_ERR_NULL;                \
        if (!(dt > 1e-6) || dt > 1.0) return FDYN_ERR_BAD_DT;                                                \
        hipLaunchKernelGGL((agent_step_kernel<S, T>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           level, x, pid_state, type, params, n_types, pid_cfg, consts, cmd, n, S(dt), n_steps, \
                           surf_out, cfg_per_lane);                                                     \
        return launch_status();                                                                              \
    }
FD_DEFINE_AGENT(fdyn_agent_step_f64, double, double)
FD_DEFINE_AGENT(fdyn_agent_step_mixed, double, float)
FD_DEFINE_AGENT(fdyn_agent_step_f32, float, float)

FD_DEFINE_CASCADE(fdyn_cascade_step_f64, double, double)
FD_DEFINE_CASCADE(fdyn_cascade_step_mixed, double, float)
FD_DEFINE_CASCADE(fdyn_cascade_step_f32, float, float)

#define FD_DEFINE_ENV(SUFFIX, S, E, T)                                                                       \
    int fdyn_rate_env_reset_##SUFFIX(S* x, E* e, int32_t* ei, float* pid_state, const uint8_t* mask,         \
                                     const double* env_consts, const double* pool, int pool_depth,           \
                                     uint64_t seed, float* obs_out, int64_t n, void* stream)                 \
    {                                                                                                        \
        FD_CHECK_COMMON(n, 1)                                                                                \
        if (pool && pool_depth < 1) return FDYN_ERR_BAD_SIZE;                                                \
        hipLaunchKernelGGL((rate_env_reset_kernel<S, E>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           x, e, ei, pid_state, mask, env_consts, pool, pool_depth, seed, obs_out, n); \
        return launch_status();                                                                              \
    }                                                                                                        \
    int fdyn_rate_env_step_##SUFFIX(S* x, E* e, int32_t* ei, const uint8_t* type, const double* params,      \
                                    int n_types, const double* env_consts, const float* actions,             \
                                    float* pid_state, const float* pid_cfg, const double* casc_consts,       \
                                    float* actions_out, const S* rw_delta, const double* pool, int pool_depth, \
                                    uint64_t seed, int auto_reset, float residual_scale, float* obs_out,     \
                                    float* reward_f32,                                                       \
                                    S* reward_full, uint8_t* terminated, uint8_t* truncated,                 \
                                    int32_t* ev_count, int32_t* ev_count_next, int32_t* ev_int,              \
                                    float* ev_flt, int ev_cap, int64_t n, void* stream)                      \
    {                                                                                                        \
        FD_CHECK_COMMON(n, n_types)                                                                          \
        if (pool && pool_depth < 1) return FDYN_ERR_BAD_SIZE;                                                \
        if ((!actions || residual_scale > 0.0f) && !(pid_state && pid_cfg && casc_consts)) return FDYN_ERR_NULL; \
        if (!obs_out || !terminated || !truncated || !env_consts) return FDYN_ERR_NULL;                      \
        if (ev_count && (!ev_int || !ev_flt)) return FDYN_ERR_NULL;                                          \
        /* records go to FD_EV_SHARDS equal segments: a capacity that is not a positive multiple would give   \
           segments of zero (or fewer than stated) records and drop episode ends silently */                  \
        if (ev_count && (ev_cap < FD_EV_SHARDS || ev_cap % FD_EV_SHARDS != 0)) return FDYN_ERR_BAD_SIZE;       \
        if (sizeof(T) == 4 && n > int64_t(simd_count()) * FD_WAVE)   /* more than one wave per SIMD */            \
            hipLaunchKernelGGL((rate_env_step_kernel<S, T, true>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           x, e, ei, type, params, n_types, env_consts, actions, pid_state, pid_cfg,         \
                           casc_consts, actions_out, rw_delta, pool, pool_depth, seed, auto_reset,           \
                           residual_scale, obs_out,                                                          \
                           reward_f32, reward_full, terminated, truncated, ev_count, ev_count_next, ev_int, ev_flt,  \
                           ev_cap, n);                                                                  \
        else                                                                                                 \
            hipLaunchKernelGGL((rate_env_step_kernel<S, T, false>), dim3(grid_for(n)), dim3(FD_BLOCK), 0, (hipStream_t)stream, \
                           x, e, ei, type, params, n_types, env_consts, actions, pid_state, pid_cfg,         \
                           casc_consts, actions_out, rw_delta, pool, pool_depth, seed, auto_reset,           \
                           residual_scale, obs_out,                                                          \
                           reward_f32, reward_full, terminated, truncated, ev_count, ev_count_next, ev_int, ev_flt,  \
                           ev_cap, n);                                                                  \
        return launch_status();                                                                              \
    }
FD_DEFINE_ENV(f64, double, double, double)
FD_DEFINE_ENV(mixed, double, float, float)
FD_DEFINE_ENV(f32, float, float, float)

}  // extern "C"
