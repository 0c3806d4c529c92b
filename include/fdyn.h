/* fdyn.h -- C-ABI of libfdyn_hip.so: the MI355X (gfx950) batched flight-dynamics hot path.
 *
 * This is the drop-in boundary beneath the reference's Python interfaces.  Plain pointers and sizes only:
 * every array argument is a DEVICE pointer (hipMalloc'd / a torch tensor's data_ptr()) unless marked "host";
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on that stream.
 * No call allocates, frees or synchronises, so every entry point can be captured into a hipGraph.
 *
 * Array layouts ("SoA" = [word][N] row-major) and all slot enums: fdyn_layout.h.
 * Precision suffixes:   _f64   state fp64, dynamics fp64  (parity variant; the reference state is float64,
 *                               simulation/simplified_6dof.py:173)
 *                       _mixed state fp64, dynamics evaluated in fp32, RK4 accumulate in fp64
 *                       _f32   state fp32, dynamics fp32   (throughput variant)
 * The PID arithmetic is fp32 in every variant, bit-faithful to cpp/src/pid_controller.cpp:24-60.
 *
 * What each entry point replaces in the reference (paths relative to its root):
 *   fdyn_sixdof_step_*     Simplified6DOF.step               simulation/simplified_6dof.py:228-293 (+ _dynamics :333-503)
 *                          looped n_sub times as             simulation/simulation_backend.py:82-101 does
 *   fdyn_num_substeps      max(1, int(dt/dt_physics))        simulation/simulation_backend.py:95
 *   fdyn_derived_*         Simplified6DOF.get_state          simulation/simplified_6dof.py:295-331,505-530
 *   fdyn_pid_compute_batch PIDController::compute            cpp/src/pid_controller.cpp:24-60 (pybind: cpp/bindings/bindings.cpp:50-62)
 *   fdyn_cascade_step_*    MissionPlanner.update + WaypointAgent/HSAAgent/AttitudeAgent/RateAgent.compute_action
 *                          + set_controls + step             examples/03_waypoint_square_demo.py:148-209,
 *                          controllers/{mission_planner.py:128-184, waypoint_agent.py:81-242, hsa_agent.py:119-231,
 *                          attitude_agent.py:86-154, rate_agent.py:65-124}
 *   fdyn_agent_step_*      {Rate,Attitude,HSA,Waypoint}Agent.compute_action (+ set_controls + step): one level commanded directly
 *   fdyn_rate_env_reset_*  RateControlEnv.reset              learned_controllers/envs/rate_env.py:151-210
 *   fdyn_rate_env_step_*   RateControlEnv.step (+ the vec-env's auto-reset, + optionally the PID demonstrator of
 *                          learned_controllers/utils/pid_demonstrations.py:47-77)
 *                                                            learned_controllers/envs/rate_env.py:212-300,342-460,
 *                                                            learned_controllers/envs/rewards.py:48-137,168-221
 *   fdyn_rate_metrics_*    MetricsCalculator.compute_metrics learned_controllers/eval/metrics.py:95-362 (fed by eval_rate.py:70-233)
 *   fdyn_rate_reward_seq_* RateTrackingReward.compute / SettlingTimeBonus.compute on recorded sequences (weights as parameters,
 *                          components out)                     learned_controllers/envs/rewards.py:48-137,168-221
 *   fdyn_sensor_update_*   NoisySensorInterface.update       interfaces/sensor.py:199-243
 *   fdyn_sensor_observe    the same noise model on RateControlEnv observations (rate_env.py:374-408 layout)
 * Policy side (the reference delegates these to torch.nn.LSTM / SB3's PPO, which are not in its tree):
 *   fdyn_policy_features, fdyn_policy_recurrent, fdyn_lstm_cell_mfma, fdyn_policy_trunks, fdyn_policy_heads, fdyn_gaussian_head, fdyn_episode_flags
 *                          rollout: features extractor / LSTM cell / trunks / output heads of
 *                          learned_controllers/networks/lstm_policy.py:13-136 (+ sb3_contrib's actor / critic LSTMs)
 *   fdyn_lstm_cell_fwd/_bwd, fdyn_lstm_seq_fwd/_bwd(_bsum),      BPTT point-wise cell update and its gradient (bias sums folded in),
 *   fdyn_lstm_cell0_fwd/_bwd, fdyn_colsum_partials                the zero-state three-gate layout of the features extractor
 *   fdyn_gae, fdyn_ppo_loss, fdyn_colsum                         GAE(lambda), clipped-surrogate loss + gradient, bias gradients
 *                          (SB3 PPO.train semantics driven by learned_controllers/train_rate.py:128-147; parity unpinned --
 *                          SB3 is absent -- and checked against the same arithmetic in plain torch)
 * The reference-side binding a maintainer would add (ctypes) is shown in INTEGRATION.md.
 */
#ifndef FDYN_H
#define FDYN_H

#include <stdint.h>
#include "fdyn_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FDYN_ABI_VERSION 2          /* 2: sharded episode-end records (FD_EV_SHARDS), fp32 env words in the mixed variant */

/* return codes: 0 ok; >0 a hipError_t from the launch; <0 argument errors */
#define FDYN_OK 0
#define FDYN_ERR_BAD_DT (-1)      /* the reference raises ValueError (simplified_6dof.py:241-245) */
#define FDYN_ERR_BAD_TYPES (-2)   /* n_types outside 1..8 */
#define FDYN_ERR_BAD_SIZE (-3)
#define FDYN_ERR_NULL (-4)

int fdyn_abi_version(void);
int fdyn_num_substeps(double dt, double dt_physics);
int fdyn_device_info(int* cu_count /*host*/, int* wave_size /*host*/, char* arch /*host*/, int arch_len);
/* smallest ev_cap with which fdyn_rate_env_step_* can never drop an episode-end record of an n-env fleet */
int64_t fdyn_event_capacity(int64_t n);

/* ---- physics ------------------------------------------------------------------------------------------------
 * x      [FD_NX][n]  state, updated in place
 * u      [FD_NU][n]  controls (elevator, aileron, rudder, throttle); clipped like set_controls
 * type   [n] uint8   aircraft-type index into params, or NULL (all type 0)
 * params [n_types][FD_NP] fp64 parameter blocks (FD_P_*), 1 <= n_types <= 8
 * advances n_sub RK4 steps of dt/n_sub each; derived_out [FD_ND][n] or NULL                                   */
int fdyn_sixdof_step_f64(double* x, const double* u, const uint8_t* type, const double* params, int n_types,
                         int64_t n, double dt, int n_sub, double* derived_out, void* stream);
int fdyn_sixdof_step_mixed(double* x, const double* u, const uint8_t* type, const double* params, int n_types,
                           int64_t n, double dt, int n_sub, double* derived_out, void* stream);
int fdyn_sixdof_step_f32(float* x, const float* u, const uint8_t* type, const double* params, int n_types,
                         int64_t n, double dt, int n_sub, float* derived_out, void* stream);
int fdyn_derived_f64(const double* x, int64_t n, double* out /*[FD_ND][n]*/, void* stream);
int fdyn_derived_f32(const float* x, int64_t n, float* out, void* stream);

/* ---- batched scalar PID --------------------------------------------------------------------------------------
 * cfg [FD_NPC] (cfg_per_lane = 0) or [n][FD_NPC] (1); state [FD_NPS][n] updated in place; out [n]             */
int fdyn_pid_compute_batch(const float* cfg, int cfg_per_lane, float* state, const float* setpoint,
                           const float* measurement, float dt, float* out, int64_t n, void* stream);

/* ---- cascade: n_steps x {mission update, waypoint->HSA->attitude->rate agents, one RK4 of dt} ---------------
 * pid_state [FD_NPID*FD_NPS][n] fp32 ; wp_idx [n] int32 (>= n_wp means mission complete)
 * pid_cfg [FD_NPID][FD_NPC] fp32 ; consts [FD_NC] fp64 (FD_C_*) ; wps [n_wp][FD_NWP] fp64, n_wp <= 16
 * surf_out [FD_NU][n] last commanded surfaces or NULL ; reached_total [n] int32 += waypoints reached, or NULL  */
int fdyn_cascade_step_f64(double* x, float* pid_state, int32_t* wp_idx, const uint8_t* type, const double* params,
                          int n_types, const float* pid_cfg, const double* consts, const double* wps, int n_wp,
                          int64_t n, double dt, int n_steps, double* surf_out, int32_t* reached_total, void* stream);
int fdyn_cascade_step_mixed(double* x, float* pid_state, int32_t* wp_idx, const uint8_t* type, const double* params,
                            int n_types, const float* pid_cfg, const double* consts, const double* wps, int n_wp,
                            int64_t n, double dt, int n_steps, double* surf_out, int32_t* reached_total, void* stream);
int fdyn_cascade_step_f32(float* x, float* pid_state, int32_t* wp_idx, const uint8_t* type, const double* params,
                          int n_types, const float* pid_cfg, const double* consts, const double* wps, int n_wp,
                          int64_t n, double dt, int n_steps, float* surf_out, int32_t* reached_total, void* stream);

/* ---- one cascade level commanded directly --------------------------------------------------------------------------
 * n_steps x {agent.compute_action(command, state, dt) -> set_controls -> one RK4 of dt}; n_steps == 0: compute_action only
 * (x untouched).  level = FD_LEVEL_* ; cmd [4][n]: RATE p, q, r, throttle | ATTITUDE roll, pitch, yaw (NaN = no yaw
 * command), throttle | HSA heading, speed, altitude, - | WAYPOINT north, east, altitude, speed (NaN = keep airspeed).
 * Replaces RateAgent / AttitudeAgent / HSAAgent / WaypointAgent.compute_action (controllers/rate_agent.py:65-124,
 * attitude_agent.py:86-154, hsa_agent.py:119-231, waypoint_agent.py:81-242) and the closed-loop helper the reference's tests
 * use (tests/test_control_integration.py:34-74).  pid_state / pid_cfg / consts as fdyn_cascade_step_*; cfg_per_lane != 0:
 * pid_cfg is [n][FD_NPID][FD_NPC], one gain set PER AIRCRAFT -- a whole gain sweep (examples/tune_pids.py tries five sets one
 * after the other) flies in one launch.                                                                                    */
int fdyn_agent_step_f64(int level, double* x, float* pid_state, const uint8_t* type, const double* params, int n_types,
                        const float* pid_cfg, int cfg_per_lane, const double* consts, const double* cmd, int64_t n, double dt,
                        int n_steps, double* surf_out, void* stream);
int fdyn_agent_step_mixed(int level, double* x, float* pid_state, const uint8_t* type, const double* params, int n_types,
                          const float* pid_cfg, int cfg_per_lane, const double* consts, const double* cmd, int64_t n, double dt,
                          int n_steps, double* surf_out, void* stream);
int fdyn_agent_step_f32(int level, float* x, float* pid_state, const uint8_t* type, const double* params, int n_types,
                        const float* pid_cfg, int cfg_per_lane, const double* consts, const float* cmd, int64_t n, double dt,
                        int n_steps, float* surf_out, void* stream);

/* ---- rate-control env ------------------------------------------------------------------------------------------
 * x [FD_NX][n] ; e [FD_NE][n] (FD_E_*; fp64 in the f64 variant, fp32 in mixed and f32 -- there FD_E_SETTLE_TIMER counts settled
 *   steps and FD_E_TIME is step * dt) ; ei [FD_NEI][n] int32 ; env_consts [FD_NEC] fp64 (FD_EC_*)
 * pool [n][pool_depth][FD_NR] fp64 host-presampled reset records (parity mode), or NULL = in-kernel Philox draws
 *      keyed by (seed, env, episode) (throughput mode)
 * reset: mask [n] uint8 (NULL = all) ; pid_state [3*FD_NPS][n] zeroed for reset envs if non-NULL ; obs_out [n][18]
 * step : actions [n][4] fp32 (aileron, elevator, rudder, throttle), or NULL => the fused rate-PID demonstrator
 *        (needs pid_state, pid_cfg [>=3][FD_NPC], casc_consts [FD_NC]); actions_out [n][4] or NULL
 *        rw_delta [3][n] this step's random-walk deltas (parity mode) or NULL
 *        residual_scale > 0 (with actions, pid_state, pid_cfg, casc_consts): ResidualRateControlEnv
 *        (learned_controllers/envs/residual_rate_env.py:99-157): action = clip(PID + scale * actions), reward += bonus
 *        auto_reset != 0: envs that end are reset in-kernel and obs_out holds the post-reset observation
 *        reward_f32 [n] / reward_full [n] (either may be NULL) ; terminated, truncated [n] uint8
 *        ev_count [FD_EV_SHARDS] int32 (must be 0 on entry), ev_int [ev_cap][FD_EV_NI], ev_flt [ev_cap][FD_EV_NF]:
 *        compacted episode-end records (env id, length, terminated | return, terminal observation) in FD_EV_SHARDS
 *        segments of ev_cap / FD_EV_SHARDS records (fdyn_layout.h); NULL = no records.  ev_cap must be a positive
 *        multiple of FD_EV_SHARDS (FDYN_ERR_BAD_SIZE otherwise); fdyn_event_capacity(n) never drops a record.
 *        ev_count_next [FD_EV_SHARDS] or NULL: a second counter set this launch clears, so two sets can be
 *        ping-ponged across steps without a memset on the stream                                              */
#define FDYN_DECLARE_ENV(SUFFIX, S, E)                                                                          \
    int fdyn_rate_env_reset_##SUFFIX(S* x, E* e, int32_t* ei, float* pid_state, const uint8_t* mask,            \
                                     const double* env_consts, const double* pool, int pool_depth,              \
                                     uint64_t seed, float* obs_out, int64_t n, void* stream);                   \
    int fdyn_rate_env_step_##SUFFIX(S* x, E* e, int32_t* ei, const uint8_t* type, const double* params,         \
                                    int n_types, const double* env_consts, const float* actions,                \
                                    float* pid_state, const float* pid_cfg, const double* casc_consts,          \
                                    float* actions_out, const S* rw_delta, const double* pool, int pool_depth,  \
                                    uint64_t seed, int auto_reset, float residual_scale, float* obs_out,        \
                                    float* reward_f32,                                                          \
                                    S* reward_full, uint8_t* terminated, uint8_t* truncated,                    \
                                    int32_t* ev_count, int32_t* ev_count_next, int32_t* ev_int,                 \
                                    float* ev_flt, int ev_cap, int64_t n, void* stream);
FDYN_DECLARE_ENV(f64, double, double)
FDYN_DECLARE_ENV(mixed, double, float)       /* fp32 env words: see fdyn_layout.h, FD_E_* */
FDYN_DECLARE_ENV(f32, float, float)

/* ---- policy-side fused kernels (csrc/policy_kernels.hip) ------------------------------------------------------------
 * LSTM cell point-wise update from pre-activation gates [B][4H] (PyTorch order i,f,g,o; bias already added by the
 * GEMM): replaces the ~40 element-wise launches torch needs per cell (nn.LSTM arithmetic used by
 * learned_controllers/networks/lstm_policy.py:49-61 and sb3_contrib's actor/critic LSTMs).
 * gates_bf16: 1 = bf16 storage for gates/h_lp/act/dh/dgates, 0 = fp32.  c_prev NULL = zero state.  H % 8 == 0.
 * fwd: h_f32 [B][H] and/or h_lp [B][H] (either may be NULL), c_out [B][H] fp32, act_out [B][4H] activated gates or NULL.
 * bwd: from act, c_prev, c_new, dh (+ dc_next or NULL) -> dgates [B][4H], dc_prev [B][H] (or NULL).  A zero-state cell
 *      (c_prev NULL) may pass c_new NULL: c = i * g is then rebuilt from the saved gates (its forward need not keep c). */
int fdyn_lstm_cell_fwd(const void* gates, int gates_bf16, const float* c_prev, float* h_f32, void* h_lp, float* c_out,
                       void* act_out, int64_t B, int H, void* stream);
int fdyn_lstm_cell_bwd(const void* act, int bf16, const float* c_prev, const float* c_new, const void* dh,
                       const float* dc_next, void* dgates, float* dc_prev, int64_t B, int H, void* stream);
/* One step of an LSTM SEQUENCE (BPTT over T inside one autograd node): the same point-wise kernels with the episode-start
 * masks and the recurrent plumbing folded in, so a step costs two launches each way (the GEMM and this).
 * fwd: c_prev is multiplied by keep [B] (NULL = 1); besides h_lp [B][H] the kernel stores h * keep_next into h_next (row
 *      stride next_stride elements: the recurrent columns of the NEXT step's [x | h] input row), or h_next = NULL;
 *      bias [B / group_rows][4H] (gates dtype) or NULL is added to the gates of rows in group row / group_rows (a batched
 *      GEMM over several cells has no bias epilogue).
 * bwd: dh_total = dh + dh2_keep * dh2 (dh2 = the recurrent columns of step t+1's input gradient, row stride dh2_stride,
 *      or NULL), summed in fp32; dgates may alias act (in place); dc_prev comes out already multiplied by keep.        */
int fdyn_lstm_seq_fwd(const void* gates, int gates_bf16, const float* c_prev, const float* keep, void* h_lp, float* c_out,
                      void* act_out, void* h_next, int64_t next_stride, const float* keep_next, const void* bias,
                      int64_t group_rows, int64_t B, int H, void* stream);
int fdyn_lstm_seq_bwd(const void* act, int bf16, const float* c_prev, const float* keep, const float* c_new, const void* dh,
                      const void* dh2, int64_t dh2_stride, const float* dh2_keep, const float* dc_next, void* dgates,
                      float* dc_prev, int64_t B, int H, void* stream);
/* The same backward step with the BIAS GRADIENT folded in: every block owns rows_per_block consecutive rows, sums its dgates
 * (the rounded values it stores) per column and writes one partial row to bias_ws [ceil(B / rows_per_block)][4H]; the partial
 * rows of all steps of a BPTT pass are then reduced by fdyn_colsum_partials instead of a second pass over [T*B][4H].
 * Needs 256 % (H / 8) == 0 and (rows_per_block * H / 8) % 256 == 0. */
int fdyn_lstm_seq_bwd_bsum(const void* act, int bf16, const float* c_prev, const float* keep, const float* c_new, const void* dh,
                           const void* dh2, int64_t dh2_stride, const float* dh2_keep, const float* dc_next, void* dgates,
                           float* dc_prev, float* bias_ws, int64_t rows_per_block, int64_t B, int H, void* stream);
/* The same backward step when the forward kept the PRE-activations instead of the activated gates (the GEMM wrote them, the
 * forward point-wise pass never re-wrote them: 8 of its 28 bytes per hidden unit): `gates` [B][4H] + bias [B / group_rows][4H]
 * (gates dtype) are activated again here; everything else as fdyn_lstm_seq_bwd(_bsum) (bias_ws may be NULL). */
int fdyn_lstm_seq_bwd_pre(const void* gates, int bf16, const void* bias, int64_t group_rows, const float* c_prev, const float* keep,
                          const float* c_new, const void* dh, const void* dh2, int64_t dh2_stride, const float* dh2_keep,
                          const float* dc_next, void* dgates, float* dc_prev, float* bias_ws, int64_t rows_per_block, int64_t B, int H,
                          void* stream);
/* Zero-state cell in the THREE-gate layout (i, g, o along 3H): the features extractor's LSTM layers, which the reference runs
 * on a length-1 sequence without carried state (learned_controllers/networks/lstm_policy.py:75-92), so the forget gate
 * multiplies zero and neither its pre-activation nor its gradient exists.  fwd: gates [B][3H] (bias added by the GEMM) ->
 * h_out [B][H], act_out [B][3H] activated gates (NULL = not kept; may alias gates).  bwd: act, dh -> dgates [B][3H] (may alias
 * act); bias_ws as above with rows of 3H floats, or NULL. */
int fdyn_lstm_cell0_fwd(const void* gates, int bf16, void* h_out, void* act_out, int64_t B, int H, void* stream);
int fdyn_lstm_cell0_bwd(const void* act, int bf16, const void* dh, void* dgates, float* bias_ws, int64_t rows_per_block,
                        int64_t B, int H, void* stream);
/* The two trunks of the policy's mlp_extractor (pi and vf: Linear(256,128)+ReLU -> Linear(128,64)+ReLU; sb3_contrib MlpLstmPolicy
 * with the reference's net_arch, learned_controllers/networks/lstm_policy.py:107-136) as ONE MFMA launch (csrc/policy_trunk.hip):
 * h_pi, h_vf [B][256] bf16 -> lat_pi, lat_vf [B][64] bf16.  W1 [2][128][256] bf16 and b1 [2][128] fp32 (index 0 = pi, 1 = vf);
 * W2p [2][64][128] bf16 with the columns of every block of 16 in the order (0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15) -- the
 * first layer's output tile is the second layer's MFMA operand as it stands -- and b2 [2][64] fp32. */
int fdyn_policy_trunks(const void* h_pi, const void* h_vf, const void* W1, const float* b1, const void* W2p, const float* b2,
                       void* lat_pi, void* lat_vf, int64_t B, void* stream);
/* The same two trunks with the output heads and the sampling behind them in ONE launch (what fdyn_policy_trunks +
 * fdyn_policy_heads do in two, with lat_pi / lat_vf never leaving the registers): Wa [4][64], ba [4], wv [64], bv [1] bf16,
 * log_std [4] fp32, Philox key (seed, row, *step) as fdyn_policy_heads -> actions [B][4], logp [B], value [B] fp32.        */
int fdyn_policy_trunks_heads(const void* h_pi, const void* h_vf, const void* W1, const float* b1, const void* W2p, const float* b2,
                             const void* Wa, const void* ba, const void* wv, const void* bv, const float* log_std, uint64_t seed,
                             const uint32_t* step, int deterministic, float* actions, float* logp, float* value, int64_t B,
                             void* stream);
/* Rollout glue in one launch: episode_start [n] = (terminated | truncated) as fp32, keep [n] = 1 - episode_start (either may be
 * NULL), *counter += 1 (NULL = none: the device-side step counter of fdyn_policy_heads / fdyn_gaussian_head's action noise). */
int fdyn_episode_flags(const uint8_t* terminated, const uint8_t* truncated, float* episode_start, float* keep, int32_t* counter,
                       int64_t n, void* stream);
/* out [N] = column sums of nb partial rows [nb][N] fp32 (two stages, no atomics); ws_mid [ceil(nb / 64)][N]. */
int fdyn_colsum_partials(const float* partial, int64_t nb, int N, float* out, float* ws_mid, void* stream);
/* Reductions of the PPO update, self-contained so that a hipGraph replay recomputes them (accumulators are cleared by a
 * kernel of the same launch sequence).  fdyn_colsum: out [N] fp32 = column sums of x [M][N] (bf16 or fp32), two stages
 * through ws [fdyn_colsum_ws_floats(...)], no atomics: bias gradients.  n_groups > 1: rows come in segments of group_rows,
 * segment j belonging to group j % n_groups (the [T][G][B] row order of a multi-cell LSTM sequence); out [n_groups][N].
 * fdyn_ppo_loss: the clipped-surrogate loss of one slice of M samples and its gradient w.r.t. the policy outputs
 * (SB3 PPO.train semantics, which learned_controllers/train_rate.py:128-147 drives): mean, actions [M][4], log_std [4],
 * values, old_logp, adv, ret [M] (old_values [M] only if clip_range_vf > 0) -> dmean [M][4], dvalues [M],
 * stats [FDYN_PPO_NSTATS] = {policy loss, value loss, approx KL, clip fraction, policy + vf_coef * value,
 * d/dlog_std[4] of the log-prob part}; ws [2] scratch.  The entropy term is added by the caller.                         */
#define FDYN_PPO_NSTATS 9
int64_t fdyn_colsum_ws_floats(int64_t M, int N, int64_t group_rows, int n_groups);   /* size of ws (floats) */
int fdyn_colsum(const void* x, int bf16, int64_t M, int N, int64_t group_rows, int n_groups, float* out, float* ws, void* stream);
int fdyn_ppo_loss(const float* mean, const float* actions, const float* log_std, const float* values, const float* old_logp,
                  const float* adv, const float* ret, const float* old_values, int normalize_adv, float clip_range,
                  float clip_range_vf, float vf_coef, int64_t M, float* dmean, float* dvalues, float* stats, float* ws,
                  void* stream);
/* The whole LSTM cell step as ONE MFMA kernel (csrc/lstm_mfma.hip): gates = [x | keep*h_prev] W^T + bias on
 * v_mfma_f32_32x32x16_bf16 with the gate non-linearity and cell update fused on the accumulators (no [B][4H] tensor).
 * x [B][kx] bf16, h_prev [B][kh] bf16, c_prev [B][H] fp32, keep [B] fp32 or NULL (0 = episode start: zero the state),
 * W [4H][kx+kh] bf16 (= [W_ih | W_hh], gate order i,f,g,o), bias [4H] fp32 (= b_ih + b_hh).
 * kh == 0: zero-state layer (the reference's feature-extractor LSTM, lstm_policy.py:75-92): no h/c input.
 * Outputs h_out [B][H] bf16, c_out [B][H] fp32 (NULL allowed), h_out_f32 [B][H] or NULL.
 * Supported (kx, kh): (128,256) (128,0) (256,0) (128,128); H % 32 == 0.                                              */
int fdyn_lstm_cell_mfma(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                        const void* W, const float* bias, void* h_out, float* c_out, float* h_out_f32,
                        int64_t B, int H, void* stream);
/* 1 = this shape may be stepped IN PLACE (h_out == h_prev, c_out == c_prev): half the state footprint, which at 65 536 rows
 * is what lets the two cells' state (201 MB) live in the 256 MB Infinity Cache from one rollout step to the next (measured:
 * rollout step 0.279 -> 0.261 ms).  Any other shape given aliased state is refused with FDYN_ERR_BAD_SIZE.                  */
int fdyn_lstm_cell_mfma_inplace_ok(int kx, int kh, int H, int64_t B);
/* Two cells of the same shape on the same x and keep -- MlpLstmPolicy's actor and critic LSTM
 * (learned_controllers/networks/lstm_policy.py:107-136, sb3_contrib's separate lstm_actor / lstm_critic) -- in ONE launch where
 * the one-wave-per-SIMD kernel applies (kx = 128, kh = H = 256, B a multiple of 256 and >= 32 768), otherwise as two calls of
 * fdyn_lstm_cell_mfma.  Operand meaning and the in-place rule as there; both cells need c_out.                               */
int fdyn_lstm_cell_mfma_pair(const void* x, int kx, const float* keep, int kh, int64_t B, int H,
                             const void* h_prev0, const float* c_prev0, const void* W0, const float* bias0, void* h_out0, float* c_out0,
                             const void* h_prev1, const float* c_prev1, const void* W1, const float* bias1, void* h_out1, float* c_out1,
                             void* stream);
/* BOTH recurrent cells of the rollout policy (sb3_contrib MlpLstmPolicy's actor and critic nn.LSTM(128, 256) over the shared
 * features; reference net sizes learned_controllers/networks/lstm_policy.py:107-136) as ONE launch with lane = batch row
 * (csrc/policy_rc64.hip).  Operands live in the kernel's own layouts (policy.py: rc_pack_x / rc_pack_h / rc_pack_c convert from
 * and to the row-major [B][..] form): feats_frag [B/64][2][8][64][8] bf16, h [B/64][2][16][64][8] bf16,
 * c [B/64][8][2][4][64][4] fp32; keep [B] fp32 or NULL; weight_image = fdyn_policy_recurrent_image_bytes() bytes
 * (policy.py: pack_rc_weights), bias [2][4H] fp32 (= b_ih + b_hh per cell, gate order i, f, g, o).  out == in updates the
 * state IN PLACE (every lane reads and writes only its own row).  B % 256 == 0.                                          */
int fdyn_policy_recurrent_image_bytes(void);
int fdyn_policy_recurrent(const void* feats_frag, const float* keep, const void* weight_image, const float* bias,
                          const void* h_pi_in, const float* c_pi_in, void* h_pi_out, float* c_pi_out,
                          const void* h_vf_in, const float* c_vf_in, void* h_vf_out, float* c_vf_out, int64_t B, void* stream);
/* BPTT forward of the same cell (csrc/lstm_mfma.hip, TRAIN instantiation): besides h_out / c_out it stores the ACTIVATED gates
 * act_out (bf16; kh > 0: [B][4H] = sigmoid(i), sigmoid(f), tanh(g), sigmoid(o); kh = 0, the zero-state layers of the features
 * extractor: [B][3H] = (i, g, o), c_out may be NULL) for fdyn_lstm_seq_bwd / fdyn_lstm_cell0_bwd, and -- h_next != NULL --
 * h' * keep_next into rows of stride next_stride (the recurrent columns of the next step's [x | h] row, which the weight-gradient
 * GEMM reads).  Replaces the training step's GEMM -> [B][4H] pre-activations in HBM -> fdyn_lstm_seq_fwd round trip. */
int fdyn_lstm_cell_mfma_train(const void* x, int kx, const void* h_prev, int kh, const float* c_prev, const float* keep,
                              const void* W, const float* bias, void* h_out, float* c_out, void* act_out,
                              void* h_next, int64_t next_stride, const float* keep_next, int64_t B, int H, void* stream);
/* The policy's features extractor as ONE kernel (csrc/policy_fe64.hip; learned_controllers/networks/lstm_policy.py:13-97):
 * obs [B][18] fp32 -> Linear(18,128)+ReLU -> two zero-state LSTM layers (128->256->256) -> Linear(256,128)+ReLU -> feats
 * [B][128] bf16, activations in registers between the layers.  weight_image: the four weight matrices as bf16 in the
 * kernel's streaming order and LDS layout (fdyn_policy_features_image_bytes() bytes; host packer: policy.pack_fe_weights),
 * bias [128 + 1024 + 1024 + 128] fp32 = embedding, b_ih + b_hh of layer 1 and 2, projection.  B % 256 == 0.               */
int fdyn_policy_features_image_bytes(void);
int fdyn_policy_features(const float* obs, const void* weight_image, const float* bias, void* feats, int64_t B, void* stream);
/* The same launch also doing the rollout glue of fdyn_episode_flags for its rows (terminated / truncated [B] uint8 of the
 * PREVIOUS env step -> episode_start [B], keep [B] fp32, either may be NULL; *counter += 1, NULL = none): the cells launched
 * behind it read keep, the heads read the counter -- one launch less per rollout step.                                      */
int fdyn_policy_features_flags(const float* obs, const void* weight_image, const float* bias, void* feats,
                               const uint8_t* terminated, const uint8_t* truncated, float* episode_start, float* keep,
                               int32_t* counter, int64_t B, void* stream);
/* Diagonal-Gaussian policy head: actions [B][4] = mean + exp(log_std) * N(0,1) (Philox keyed by seed, env, *step -- a
 * uint32 counter in DEVICE memory the caller increments on the stream, so graph replays draw fresh noise; or the mean
 * itself when deterministic), logp [B] = log-probability of the sampled action.  mean [B][4] bf16 (mean_bf16=1) or fp32. */
int fdyn_gaussian_head(const void* mean, int mean_bf16, const float* log_std, uint64_t seed, const uint32_t* step, int deterministic,
                       float* actions, float* logp, int64_t B, void* stream);
/* Both output heads + sampling in one launch: mean = pi_hidden [B][64] Wa^T + ba (action_net 64 -> 4), value = vf_hidden
 * [B][64] wv + bv (value_net 64 -> 1), then the Gaussian head.  All weights / hiddens bf16; actions [B][4], logp, value fp32. */
int fdyn_policy_heads(const void* pi_hidden, const void* vf_hidden, const void* Wa, const void* ba, const void* wv, const void* bv,
                      const float* log_std, uint64_t seed, const uint32_t* step, int deterministic, float* actions, float* logp,
                      float* value, int64_t B, void* stream);
/* GAE(lambda) over a [T][N] rollout (one lane per env): adv, ret [T][N].  episode_starts[t][n] = 1 if env n was reset
 * before step t; last_values / last_dones [N] describe the state after the final step.                              */
int fdyn_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
             const float* last_dones, float gamma, float lam, int T, int64_t N, float* adv, float* ret, void* stream);


/* ---- evaluation metrics (csrc/eval_kernels.hip) -----------------------------------------------------------------------
 * MetricsCalculator.compute_metrics (learned_controllers/eval/metrics.py:95-362) for n recorded episodes at once.
 * times [T] fp64 shared time base (times[t] = env time after step t+1, as eval_rate.py:104,219 records info["time"]);
 * rates, commands [T][3][n]; actions [T][n][4] fp32; rewards [T][n]; lengths [n] int32 = steps of each episode (<= T);
 * settle_steps = int(settling_duration / dt).  out [FD_NM][n] fp64 in RateControlMetrics field order (FD_M_*; success
 * as 0/1).  An axis whose |command| stays < 0.01 keeps zeros (metrics.py:144-145); a zero-length episode gives zeros. */
int fdyn_rate_metrics_f64(const double* times, const double* rates, const double* commands, const float* actions,
                          const double* rewards, const int32_t* lengths, double settling_threshold, int settle_steps,
                          int T, int64_t n, double* out, void* stream);
int fdyn_rate_metrics_f32(const double* times, const float* rates, const float* commands, const float* actions,
                          const float* rewards, const int32_t* lengths, double settling_threshold, int settle_steps,
                          int T, int64_t n, double* out, void* stream);


/* ---- rewards on recorded data (csrc/eval_kernels.hip) -----------------------------------------------------------------
 * RateTrackingReward.compute + SettlingTimeBonus.compute (learned_controllers/envs/rewards.py:48-137,168-221) over n
 * independent sequences of T steps, one sequence per lane -- the same arithmetic the fused env step applies
 * (rate_env.py:248-279), here with the reference's configurable weights and with the per-step components the env step
 * does not materialise.  errs [T][3][n]; actions [T][4][n] ([aileron, elevator, rudder, throttle]); prev0 [4][n] = the
 * action before step 0 (later steps use the previous row); flight [T][FD_NRF][n]; cmd [3][n]; params [FD_NRW] fp64;
 * rstate [FD_NRS][n] in/out (zeros = reset()).  Outputs (any may be NULL): tracking [T][n] (the RateTrackingReward total),
 * components [T][FD_NRC][n], settle [T][n] (the bonus of each step), settled [T][n] uint8 (is_settled after the step). */
int fdyn_rate_reward_seq_f64(const double* errs, const double* actions, const double* prev0, const double* flight,
                             const double* cmd, const double* params, double* rstate, double dt, int T, int64_t n,
                             double* tracking, double* components, double* settle, uint8_t* settled, void* stream);
int fdyn_rate_reward_seq_f32(const float* errs, const float* actions, const float* prev0, const float* flight,
                             const float* cmd, const double* params, float* rstate, float dt, int T, int64_t n,
                             float* tracking, float* components, float* settle, uint8_t* settled, void* stream);


/* ---- sensor layer (csrc/sensor_kernels.hip) ---------------------------------------------------------------------------
 * NoisySensorInterface.update (interfaces/sensor.py:199-243) for n aircraft: meas [FD_NMS][n] = the 12 state words +
 * airspeed + altitude with Gaussian noise, body rates additionally offset by the gyro bias; bias [FD_NSB][n]
 * (gyro 3 | accel 3) random-walks in place (sensor.reset() = zero it).  x [FD_NX][n]; derived [FD_ND][n] (the backend's
 * airspeed / altitude rows) or NULL = computed from x; noise_cfg [FD_NSN] fp64 (FD_SN_*; FD_SN_ENABLED = 0 copies the
 * truth through).  z [FD_NSZ][n] = this update's standard normals in the reference's draw order (parity mode), or NULL =
 * in-kernel Philox keyed by (seed, aircraft, *step); step = uint32 in device memory (NULL = 0).                          */
int fdyn_sensor_update_f64(const double* x, const double* derived, double* bias, const double* noise_cfg, const double* z,
                           uint64_t seed, const uint32_t* step, double* meas, int64_t n, void* stream);
int fdyn_sensor_update_f32(const float* x, const float* derived, float* bias, const double* noise_cfg, const float* z,
                           uint64_t seed, const uint32_t* step, float* meas, int64_t n, void* stream);
/* The same model applied in place to rate-control observations obs [n][18] (rate_env.py:374-408): rates += noise + gyro
 * bias, rate errors recomputed from the measured rates, airspeed / altitude / attitude += noise; gyro_bias [3][n] fp32
 * walks in place and restarts from zero where reset_mask [n] (NULL = never) is set (first observation of an episode).  */
int fdyn_sensor_observe(float* obs, float* gyro_bias, const uint8_t* reset_mask, const double* noise_cfg, const float* z,
                        uint64_t seed, const uint32_t* step, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDYN_H */
