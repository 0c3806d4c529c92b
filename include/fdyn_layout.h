/* fdyn_layout.h -- flat-array layouts shared by the C-ABI (include/fdyn.h), the host mirror and the oracle.
 *
 * Everything that crosses the boundary is a plain array of scalars; these enums name the slots.
 * Reference sources the slots come from are cited per block (paths relative to the reference root).
 */
#ifndef FDYN_LAYOUT_H
#define FDYN_LAYOUT_H

/* ---- 12-word rigid-body state, simulation/simplified_6dof.py:172-173 ------------------------------- */
enum {
    FD_X_N = 0, FD_X_E, FD_X_D,        /* position NED (m)            */
    FD_X_U, FD_X_V, FD_X_W,            /* body velocity (m/s)         */
    FD_X_ROLL, FD_X_PITCH, FD_X_YAW,   /* Euler angles (rad)          */
    FD_X_P, FD_X_Q, FD_X_R,            /* body rates (rad/s)          */
    FD_NX = 12
};

/* ---- 4 control words, controllers/types.py:173-201 (ControlSurfaces.to_array order) ---------------- */
enum { FD_U_ELEVATOR = 0, FD_U_AILERON, FD_U_RUDDER, FD_U_THROTTLE, FD_NU = 4 };

/* ---- 4 derived words, simplified_6dof.py:295-331 (get_state) --------------------------------------- */
enum { FD_D_AIRSPEED = 0, FD_D_ALTITUDE, FD_D_GROUND_SPEED, FD_D_HEADING, FD_ND = 4 };

/* ---- aircraft parameter block (one per aircraft TYPE), simplified_6dof.py:31-117,181-187 ------------
 * The reference has no aero tables: this <=64-word block IS the "coefficient table"; the kernels stage
 * all n_types blocks in LDS and each lane picks its type's block.                                      */
enum {
    FD_P_MASS = 0, FD_P_IXX, FD_P_IYY, FD_P_IZZ,
    FD_P_WING_AREA, FD_P_WING_SPAN, FD_P_CHORD,
    FD_P_CL_0, FD_P_CL_ALPHA, FD_P_CD_0, FD_P_CD_ALPHA2,
    FD_P_CL_ELEVATOR, FD_P_CM_ELEVATOR, FD_P_CY_RUDDER, FD_P_CN_RUDDER, FD_P_CL_AILERON,
    FD_P_CM_ALPHA, FD_P_CN_BETA, FD_P_CL_BETA,
    FD_P_DAMPING_ROLL, FD_P_DAMPING_PITCH, FD_P_DAMPING_YAW,
    FD_P_MAX_THRUST, FD_P_AIR_DENSITY, FD_P_GRAVITY,
    FD_P_MIN_AIRSPEED_AERO, FD_P_MIN_U_VELOCITY,
    FD_P_MAX_ELEVATOR_RAD, FD_P_MAX_AILERON_RAD, FD_P_MAX_RUDDER_RAD,   /* np.radians(deg) done on host */
    FD_P_THRUST_ZERO_VELOCITY,
    FD_P_MAX_VELOCITY, FD_P_MAX_RATE_RAD, FD_P_MAX_PITCH_RAD, FD_P_MAX_ALPHA_RAD,
    FD_P_MAX_ACCELERATION, FD_P_MAX_ANGULAR_ACCELERATION,
    FD_P_MAX_TIMESTEP, FD_P_MIN_TIMESTEP,
    FD_NP_USED,
    FD_NP = 48,                        /* padded block stride (words) of the caller's array */
    /* derived words: exist only in the kernels' OWN staged (LDS) copy of a block, whose stride is FD_NP_STAGED; callers
     * never see them */
    FD_PD_INV_MASS = FD_NP_USED, FD_PD_INV_IXX, FD_PD_INV_IYY, FD_PD_INV_IZZ, FD_PD_SIN_MAX_ALPHA, FD_PD_COS_MAX_ALPHA,
    FD_PD_INV_THRUST_ZERO_V, FD_PD_TAN_ALPHA_FAST, FD_PD_ALPHA_NEEDS_ATAN2, FD_PD_SIN_MAX_PITCH, FD_PD_COS_MAX_PITCH,
    FD_NP_STAGED = 52
};

/* ---- scalar PID, cpp/include/pid_controller.h:22-49 and cpp/src/pid_controller.cpp:24-60 ----------- */
enum { FD_PC_KP = 0, FD_PC_KI, FD_PC_KD, FD_PC_OUT_MIN, FD_PC_OUT_MAX, FD_PC_INT_MIN, FD_PC_INT_MAX,
       FD_PC_ALPHA, FD_NPC = 8 };                      /* config: 8 x f32 */
enum { FD_PS_INTEGRAL = 0, FD_PS_ERR_PREV, FD_PS_DFILT, FD_NPS = 3 };   /* carried state: 3 x f32 */

/* the nine PIDs of the cascade, innermost first (rate_agent.py:41-50, attitude_agent.py:43-63,
 * hsa_agent.py:57-95)                                                                                  */
enum {
    FD_PID_RATE_ROLL = 0, FD_PID_RATE_PITCH, FD_PID_RATE_YAW,
    FD_PID_ATT_ROLL, FD_PID_ATT_PITCH, FD_PID_ATT_YAW,
    FD_PID_HEADING, FD_PID_ENERGY, FD_PID_BALANCE,
    FD_NPID = 9
};

/* ---- cascade glue constants (fp64 on the host; narrowed by the f32 kernels) -------------------------
 * rate_agent.py:52-55, attitude_agent.py:68-76, hsa_agent.py:101-109, waypoint_agent.py:50-65,
 * controllers/config_loader.py:40-85, controllers/mission_planner.py:173-184                           */
enum {
    FD_C_MAX_ROLL_RATE = 0, FD_C_MAX_PITCH_RATE, FD_C_MAX_YAW_RATE,   /* rad/s */
    FD_C_MAX_ROLL, FD_C_MAX_PITCH,                                    /* rad   */
    FD_C_MAX_BANK_RAD, FD_C_BASELINE_THROTTLE, FD_C_LOAD_FACTOR_GAIN, FD_C_MAX_PITCH_CMD_RAD,
    FD_C_GUIDANCE_TYPE,                                               /* 0 LOS, 1 PP, 2 default */
    FD_C_WP_MAX_BANK_RAD, FD_C_LOS_MAX_BANK_RAD, FD_C_LOS_LEAD_ANGLE_RAD,
    FD_C_LOOKAHEAD_TIME, FD_C_LOOKAHEAD_MIN, FD_C_LOOKAHEAD_MAX, FD_C_PROXIMITY_SCALE,
    FD_C_TURN_THRESHOLD_DIST, FD_C_TURN_THRESHOLD_ANGLE_RAD, FD_C_MAX_SPEED_REDUCTION, FD_C_MIN_SPEED,
    FD_C_ACCEPTANCE_RADIUS,
    FD_C_ON_COMPLETE,                                                 /* 0 freeze, 1 restart mission */
    /* fused rate-PID driver of the env kernels: throttle it holds (0.6 in pid_demonstrations.py:62 and
     * residual_rate_env.py:113, 0.5 in eval_rate.py:196) and the dt it hands the PIDs (0 = the env dt, as
     * pid_demonstrations.py:66; eval_rate.py:200 passes none => ControllerConfig.rate_loop_dt, types.py:342) */
    FD_C_PID_THROTTLE, FD_C_PID_DT,
    FD_NC = 28,
    /* derived, in the kernels' staged copy only: 1 / (g tan(bank limit)) of the two guidance laws (waypoint_agent.py:121,148) */
    FD_CD_WP_INV_G_TAN_BANK = FD_NC, FD_CD_LOS_INV_G_TAN_BANK,
    FD_NC_STAGED = 30
};
enum { FD_GUIDANCE_LOS = 0, FD_GUIDANCE_PP = 1, FD_GUIDANCE_DEFAULT = 2 };
/* control levels, numbered as the reference's ControlMode (controllers/types.py:13-24): the level a command enters at */
enum { FD_LEVEL_WAYPOINT = 1, FD_LEVEL_HSA = 2, FD_LEVEL_ATTITUDE = 3, FD_LEVEL_RATE = 4 };
enum { FD_WP_NORTH = 0, FD_WP_EAST, FD_WP_ALTITUDE, FD_WP_SPEED, FD_NWP = 4 };  /* waypoint row */
#define FD_MAX_WAYPOINTS 16

/* ---- rate-control env, learned_controllers/envs/rate_env.py ------------------------------------------ */
enum { FD_OBS_DIM = 18, FD_ACT_DIM = 4 };              /* rate_env.py:107-138, action = [ail, elev, rud, thr] */
enum { FD_CMD_STEP = 0, FD_CMD_RAMP = 1, FD_CMD_SINE = 2, FD_CMD_RANDOM_WALK = 3 };   /* rate_env.py:302-372 */

/* per-env carried words (one SoA row each, length N) */
enum {
    FD_E_CMD_P = 0, FD_E_CMD_Q, FD_E_CMD_R,            /* rate_command                         */
    FD_E_PREV_AIL, FD_E_PREV_ELEV, FD_E_PREV_RUD, FD_E_PREV_THR,   /* prev_action          */
    FD_E_PERR_P, FD_E_PERR_Q, FD_E_PERR_R,             /* RateTrackingReward.prev_errors       */
    FD_E_SIGN_P, FD_E_SIGN_Q, FD_E_SIGN_R,             /* RateTrackingReward.sign_changes      */
    FD_E_SETTLE_TIMER, FD_E_IS_SETTLED,                /* SettlingTimeBonus                    */
    FD_E_TIME,                                         /* current_time (+= dt each step)       */
    FD_E_SCHED0, FD_E_SCHED1, FD_E_SCHED2,             /* ramp end / sine amplitudes           */
    FD_E_SCHED3,                                       /* sine frequency                       */
    FD_E_EP_RETURN,                                    /* Monitor-style running episode return */
    FD_NE = 21
};
/* per-env integer words */
enum { FD_EI_STEP = 0, FD_EI_EPISODE, FD_NEI = 2 };

/* env constants (fp64) */
enum {
    FD_EC_DT = 0, FD_EC_DT_PHYSICS, FD_EC_MAX_STEPS, FD_EC_CMD_TYPE, FD_EC_DIFFICULTY_SCALE,
    FD_EC_MAX_RATE_P, FD_EC_MAX_RATE_Q, FD_EC_MAX_RATE_R,
    FD_NEC = 8
};

/* one pre-sampled reset record (host NumPy pools in parity mode): 8 IC words + 7 command words        */
enum {
    FD_R_AIRSPEED = 0, FD_R_ALTITUDE, FD_R_ROLL, FD_R_PITCH, FD_R_YAW, FD_R_P, FD_R_Q, FD_R_R,
    FD_R_CMD0, FD_R_CMD1, FD_R_CMD2,                   /* step: command; ramp: end; sine: amplitudes */
    FD_R_CMD3,                                         /* sine: frequency                            */
    FD_NR = 12
};

/* one compacted episode-end record (written by the wave-ballot compaction)                            */
enum { FD_EV_ENV = 0, FD_EV_LENGTH, FD_EV_TERMINATED, FD_EV_NI = 3 };           /* int32 part      */
/* float part: [0] = episode return, [1..18] = terminal observation                                  */
#define FD_EV_NF (1 + FD_OBS_DIM)
/* The record list is kept in FD_EV_SHARDS segments, each with its own counter: workgroup b appends to shard b % FD_EV_SHARDS.
 * One counter for the whole fleet is ONE memory-side atomic word -- it saturates near 88 returning atomics per microsecond
 * on MI355X, and a step in which most waves hold a finished episode (random actions: ~1000 waves) spent 11 us of its 51 us
 * queueing on it (rocprofv3 SQ_WAIT_ANY, round 2).  Shard s owns records [s * cap_s, s * cap_s + count[s]) with
 * cap_s = ev_cap / FD_EV_SHARDS; size ev_cap with fdyn_event_capacity(n) to never lose a record.                  */
enum { FD_EV_SHARDS = 64 };

/* ---- per-episode evaluation metrics, learned_controllers/eval/metrics.py:8-40 (field order of RateControlMetrics) */
enum {
    FD_M_SETTLE_ROLL = 0, FD_M_SETTLE_PITCH, FD_M_SETTLE_YAW,          /* s                           */
    FD_M_OVERSHOOT_ROLL, FD_M_OVERSHOOT_PITCH, FD_M_OVERSHOOT_YAW,     /* %                           */
    FD_M_SSERR_ROLL, FD_M_SSERR_PITCH, FD_M_SSERR_YAW,                 /* rad/s                       */
    FD_M_RISE_ROLL, FD_M_RISE_PITCH, FD_M_RISE_YAW,                    /* s                           */
    FD_M_SMOOTHNESS, FD_M_RMSE, FD_M_SUCCESS, FD_M_EPISODE_LENGTH, FD_M_TOTAL_REWARD,
    FD_NM = 17
};

/* ---- stand-alone reward evaluation (fdyn_rate_reward_seq_*), learned_controllers/envs/rewards.py ------------------- */
/* parameters (fp64): RateTrackingReward weights :14-19, then SettlingTimeBonus :160-162                              */
enum {
    FD_RW_TRACKING = 0, FD_RW_SMOOTHNESS, FD_RW_STABILITY, FD_RW_OSCILLATION, FD_RW_SURVIVAL,
    FD_RW_SETTLE_THRESHOLD, FD_RW_MIN_SETTLE_TIME, FD_RW_BONUS_MULTIPLIER,
    FD_NRW = 8
};
/* carried state per sequence: RateTrackingReward.prev_errors / sign_changes (:37-38), SettlingTimeBonus timer / flag */
enum {
    FD_RS_PERR_P = 0, FD_RS_PERR_Q, FD_RS_PERR_R, FD_RS_SIGN_P, FD_RS_SIGN_Q, FD_RS_SIGN_R, FD_RS_SETTLE_TIMER, FD_RS_IS_SETTLED,
    FD_NRS = 8
};
/* reward components in the order of the reference's dict (:125-131); flight rows of the input                        */
enum { FD_RC_TRACKING = 0, FD_RC_SMOOTHNESS, FD_RC_STABILITY, FD_RC_OSCILLATION, FD_RC_SURVIVAL, FD_NRC = 5 };
enum { FD_RF_AIRSPEED = 0, FD_RF_ALTITUDE, FD_RF_ROLL, FD_RF_PITCH, FD_NRF = 4 };

/* ---- sensor layer, interfaces/sensor.py:137-243 (NoisySensorInterface) ------------------------------------------- */
/* noise configuration (fp64): standard deviations in the order the reference draws (:208-235), then the two bias
 * random-walk steps it hard-codes (:233-234) and the enabled flag (:203-205)                                        */
enum {
    FD_SN_GPS_POS = 0, FD_SN_GPS_VEL, FD_SN_ATTITUDE, FD_SN_GYRO, FD_SN_AIRSPEED, FD_SN_ALTITUDE,
    FD_SN_GYRO_BIAS_WALK, FD_SN_ACCEL_BIAS_WALK, FD_SN_ENABLED,
    FD_NSN = 12
};
/* one update consumes 20 standard normals, in the reference's call order                                           */
enum {
    FD_SZ_POS = 0, FD_SZ_VEL = 3, FD_SZ_ATT = 6, FD_SZ_GYRO = 9, FD_SZ_AIRSPEED = 12, FD_SZ_ALTITUDE = 13,
    FD_SZ_GYRO_BIAS = 14, FD_SZ_ACCEL_BIAS = 17, FD_NSZ = 20
};
/* measurement block: the 12 state words (FD_X_* order) + airspeed + altitude ; bias block: gyro(3) + accel(3)       */
enum { FD_MS_AIRSPEED = 12, FD_MS_ALTITUDE = 13, FD_NMS = 14, FD_NSB = 6 };

#endif /* FDYN_LAYOUT_H */
