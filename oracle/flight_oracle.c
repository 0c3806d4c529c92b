/* flight_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see flight_oracle.h).
 *
 * Plain-C restatement of the reference hot path: fp64 physics and glue, fp32 PID, operation order kept as
 * the reference's Python/C++ expressions evaluate (left-associative), compiled with -ffp-contract=off.
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 */
#include "flight_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------- small helpers with the reference's Python / NumPy semantics ---------------------------- */
static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); } /* np.clip: NaN stays */
static inline double pymax(double a, double b) { return b > a ? b : a; }   /* Python max(a, b) */
static inline double pymin(double a, double b) { return b < a ? b : a; }   /* Python min(a, b) */
static inline double signd(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : (x == 0.0 ? 0.0 : x)); } /* np.sign */
static inline double deg2rad(double d) { return d * (M_PI / 180.0); }      /* np.radians */
static inline double wrap_atan2(double a) { return atan2(sin(a), cos(a)); }

double orc_wrap_angle(double a)
{   /* controllers/utils/pid_utils.py:65  (angle + pi) % (2 pi) - pi, Python floor-mod */
    const double b = 2.0 * M_PI;
    double x = a + M_PI;
    double m = fmod(x, b);
    if (m != 0.0) { if (m < 0.0) m += b; } else { m = copysign(0.0, b); }
    return m - M_PI;
}

/* ---------- parameters ------------------------------------------------------------------------------ */
void orc_params_default(double P[FD_NP], int aircraft_type)
{   /* simulation/simplified_6dof.py:45-117 ; radians pre-computed as in :182-187 */
    memset(P, 0, sizeof(double) * FD_NP);
    P[FD_P_MASS] = 8.0; P[FD_P_IXX] = 0.4; P[FD_P_IYY] = 0.6; P[FD_P_IZZ] = 0.7;
    P[FD_P_WING_AREA] = 0.5; P[FD_P_WING_SPAN] = 2.0; P[FD_P_CHORD] = 0.25;
    P[FD_P_CL_0] = 0.4; P[FD_P_CL_ALPHA] = 5.0; P[FD_P_CD_0] = 0.025; P[FD_P_CD_ALPHA2] = 0.04;
    P[FD_P_CL_ELEVATOR] = 0.6; P[FD_P_CM_ELEVATOR] = -2.0; P[FD_P_CY_RUDDER] = 0.5;
    P[FD_P_CN_RUDDER] = -0.40; P[FD_P_CL_AILERON] = 0.50;
    P[FD_P_CM_ALPHA] = -0.15; P[FD_P_CN_BETA] = 0.04; P[FD_P_CL_BETA] = -0.02;
    P[FD_P_DAMPING_ROLL] = -0.8; P[FD_P_DAMPING_PITCH] = -2.0; P[FD_P_DAMPING_YAW] = -0.6;
    P[FD_P_MAX_THRUST] = 50.0; P[FD_P_AIR_DENSITY] = 1.225; P[FD_P_GRAVITY] = 9.81;
    P[FD_P_MIN_AIRSPEED_AERO] = 10.0; P[FD_P_MIN_U_VELOCITY] = 0.1;
    P[FD_P_MAX_ELEVATOR_RAD] = deg2rad(30.0); P[FD_P_MAX_AILERON_RAD] = deg2rad(30.0);
    P[FD_P_MAX_RUDDER_RAD] = deg2rad(30.0);
    P[FD_P_THRUST_ZERO_VELOCITY] = 50.0;
    P[FD_P_MAX_VELOCITY] = 100.0; P[FD_P_MAX_RATE_RAD] = deg2rad(360.0);
    P[FD_P_MAX_PITCH_RAD] = deg2rad(85.0); P[FD_P_MAX_ALPHA_RAD] = deg2rad(30.0);
    P[FD_P_MAX_ACCELERATION] = 50.0; P[FD_P_MAX_ANGULAR_ACCELERATION] = 1000.0;
    P[FD_P_MAX_TIMESTEP] = 1.0; P[FD_P_MIN_TIMESTEP] = 1e-6;
    if (aircraft_type == 1) {   /* simulation/simulation_backend.py:64-74 'cessna' */
        P[FD_P_MASS] = 15.0; P[FD_P_IXX] = 1.0; P[FD_P_IYY] = 2.0; P[FD_P_IZZ] = 2.5;
        P[FD_P_WING_AREA] = 1.0; P[FD_P_WING_SPAN] = 3.0; P[FD_P_MAX_THRUST] = 80.0;
    }
}

/* ---------- physics --------------------------------------------------------------------------------- */
void orc_dynamics(const double *P, const double x[FD_NX], const double c[FD_NU], double xd[FD_NX])
{   /* simulation/simplified_6dof.py:333-503 */
    const double u = x[3], v = x[4], w = x[5];
    const double phi = x[6], theta = x[7], psi = x[8];
    const double p = x[9], q = x[10], r = x[11];
    const double sin_phi = sin(phi), cos_phi = cos(phi);
    const double sin_theta = sin(theta), cos_theta = cos(theta);
    const double sin_psi = sin(psi), cos_psi = cos(psi);

    const double airspeed = sqrt(u * u + v * v + w * w);                       /* :363 */
    const double safe_airspeed = pymax(airspeed, P[FD_P_MIN_AIRSPEED_AERO]);   /* :364 */
    const double min_u = P[FD_P_MIN_U_VELOCITY];
    const double u_safe = fabs(u) > 1e-6 ? pymax(fabs(u), min_u) * signd(u) : min_u;   /* :368 */
    double alpha = atan2(w, u_safe);
    alpha = clipd(alpha, -P[FD_P_MAX_ALPHA_RAD], P[FD_P_MAX_ALPHA_RAD]);
    const double sin_alpha = sin(alpha), cos_alpha = cos(alpha);
    const double beta = asin(clipd(v / safe_airspeed, -1.0, 1.0));             /* :376 */
    const double q_dyn = 0.5 * P[FD_P_AIR_DENSITY] * (airspeed * airspeed);    /* :379 (unclamped V) */

    const double elevator_rad = c[FD_U_ELEVATOR] * P[FD_P_MAX_ELEVATOR_RAD];
    const double cl = P[FD_P_CL_0] + P[FD_P_CL_ALPHA] * alpha + P[FD_P_CL_ELEVATOR] * elevator_rad;
    const double cd = P[FD_P_CD_0] + P[FD_P_CD_ALPHA2] * (alpha * alpha);
    const double rudder_rad = c[FD_U_RUDDER] * P[FD_P_MAX_RUDDER_RAD];
    const double cy = P[FD_P_CY_RUDDER] * rudder_rad;
    const double q_S = q_dyn * P[FD_P_WING_AREA];
    const double lift = q_S * cl, drag = q_S * cd, side_force = q_S * cy;
    const double fx_aero = -drag * cos_alpha + lift * sin_alpha;               /* :397-399 */
    const double fy_aero = side_force;
    const double fz_aero = -drag * sin_alpha - lift * cos_alpha;

    const double thrust_factor = pymax(0.0, 1.0 - airspeed / P[FD_P_THRUST_ZERO_VELOCITY]);   /* :403 */
    const double thrust = P[FD_P_MAX_THRUST] * c[FD_U_THROTTLE] * thrust_factor;

    const double g = P[FD_P_GRAVITY], mass = P[FD_P_MASS];
    const double fx = fx_aero + thrust + (-g * sin_theta) * mass;              /* :409-411 */
    const double fy = fy_aero + (g * cos_theta * sin_phi) * mass;
    const double fz = fz_aero + (g * cos_theta * cos_phi) * mass;

    const double aileron_rad = c[FD_U_AILERON] * P[FD_P_MAX_AILERON_RAD];
    const double b = P[FD_P_WING_SPAN], ch = P[FD_P_CHORD];
    const double half_span_over_V = b / (2 * safe_airspeed);
    const double half_chord_over_V = ch / (2 * safe_airspeed);
    const double l_moment = q_S * b * (P[FD_P_CL_AILERON] * aileron_rad
                                       + P[FD_P_DAMPING_ROLL] * p * half_span_over_V
                                       + P[FD_P_CL_BETA] * beta);              /* :420-423 */
    const double m_moment = q_S * ch * (P[FD_P_CM_ELEVATOR] * elevator_rad
                                        + P[FD_P_CM_ALPHA] * alpha
                                        + P[FD_P_DAMPING_PITCH] * q * half_chord_over_V);
    const double n_moment = q_S * b * (P[FD_P_CN_RUDDER] * rudder_rad
                                       + P[FD_P_DAMPING_YAW] * r * half_span_over_V
                                       + P[FD_P_CN_BETA] * beta);

    const double sps = sin_phi * sin_theta, cps = cos_phi * sin_theta;         /* :440-452 */
    xd[0] = cos_theta * cos_psi * u + (sps * cos_psi - cos_phi * sin_psi) * v
            + (cps * cos_psi + sin_phi * sin_psi) * w;
    xd[1] = cos_theta * sin_psi * u + (sps * sin_psi + cos_phi * cos_psi) * v
            + (cps * sin_psi - sin_phi * cos_psi) * w;
    xd[2] = -sin_theta * u + sin_phi * cos_theta * v + cos_phi * cos_theta * w;

    const double inv_mass = 1.0 / mass;                                        /* :455-460 */
    xd[3] = fx * inv_mass - q * w + r * v;
    xd[4] = fy * inv_mass - r * u + p * w;
    xd[5] = fz * inv_mass - p * v + q * u;

    const double theta_safe = clipd(theta, -P[FD_P_MAX_PITCH_RAD], P[FD_P_MAX_PITCH_RAD]);   /* :463-471 */
    const double cos_ts = cos(theta_safe), tan_ts = tan(theta_safe);
    xd[6] = p + sin_phi * tan_ts * q + cos_phi * tan_ts * r;
    xd[7] = cos_phi * q - sin_phi * r;
    xd[8] = (sin_phi * q + cos_phi * r) / cos_ts;

    const double Ixx = P[FD_P_IXX], Iyy = P[FD_P_IYY], Izz = P[FD_P_IZZ];      /* :474-482 */
    xd[9] = (l_moment - (Izz - Iyy) * q * r) / Ixx;
    xd[10] = (m_moment - (Ixx - Izz) * p * r) / Iyy;
    xd[11] = (n_moment - (Iyy - Ixx) * p * q) / Izz;

    const double maa = P[FD_P_MAX_ANGULAR_ACCELERATION], ma = P[FD_P_MAX_ACCELERATION];   /* :485-490 */
    for (int i = 9; i < 12; ++i) xd[i] = clipd(xd[i], -maa, maa);
    for (int i = 3; i < 6; ++i) xd[i] = clipd(xd[i], -ma, ma);

    int finite = 1;                                                            /* :496-501 */
    for (int i = 0; i < 12; ++i) finite &= isfinite(xd[i]) ? 1 : 0;
    if (!finite) for (int i = 0; i < 12; ++i) if (!isfinite(xd[i])) xd[i] = 0.0;
}

int orc_rk4_step(const double *P, double x[FD_NX], const double c[FD_NU], double dt)
{   /* simulation/simplified_6dof.py:228-293 */
    if (dt <= P[FD_P_MIN_TIMESTEP] || dt > P[FD_P_MAX_TIMESTEP]) return -1;    /* :241-245 ValueError */
    double k1[12], k2[12], k3[12], k4[12], xt[12];
    orc_dynamics(P, x, c, k1);
    for (int i = 0; i < 12; ++i) xt[i] = x[i] + 0.5 * dt * k1[i];
    orc_dynamics(P, xt, c, k2);
    for (int i = 0; i < 12; ++i) xt[i] = x[i] + 0.5 * dt * k2[i];
    orc_dynamics(P, xt, c, k3);
    for (int i = 0; i < 12; ++i) xt[i] = x[i] + dt * k3[i];
    orc_dynamics(P, xt, c, k4);
    for (int i = 0; i < 12; ++i) x[i] = x[i] + (dt / 6.0) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);   /* :253 */

    for (int i = 0; i < 3; ++i) {                                              /* :258 nan_to_num */
        if (isnan(x[i])) x[i] = 0.0;
        else if (isinf(x[i])) x[i] = x[i] > 0 ? 10000.0 : -10000.0;
    }
    const double mv = P[FD_P_MAX_VELOCITY];
    for (int i = 3; i < 6; ++i) x[i] = clipd(x[i], -mv, mv);                   /* :262 */
    x[6] = atan2(sin(x[6]), cos(x[6]));                                        /* :266 */
    x[7] = clipd(x[7], -P[FD_P_MAX_PITCH_RAD], P[FD_P_MAX_PITCH_RAD]);         /* :268 */
    x[8] = atan2(sin(x[8]), cos(x[8]));                                        /* :270 */
    const double mr = P[FD_P_MAX_RATE_RAD];
    for (int i = 9; i < 12; ++i) x[i] = clipd(x[i], -mr, mr);                  /* :273 */
    if (-x[2] < 0) {                                                           /* :276-283 */
        x[2] = 0.0;
        x[5] = x[5] > 0.0 ? x[5] : 0.0;                                        /* max(0.0, w) */
    }
    int finite = 1;                                                            /* :286-291 */
    for (int i = 0; i < 12; ++i) finite &= isfinite(x[i]) ? 1 : 0;
    if (!finite) for (int i = 0; i < 12; ++i) if (!isfinite(x[i])) x[i] = 0.0;
    return 0;
}

int orc_num_substeps(double dt, double dt_physics)
{   /* simulation/simulation_backend.py:95  max(1, int(dt / dt_physics)) */
    double r = dt / dt_physics;
    long n = (long)r;   /* int() truncates toward zero */
    return n < 1 ? 1 : (int)n;
}

int orc_backend_step(const double *P, double x[FD_NX], const double c[FD_NU], double dt, double dt_physics)
{   /* simulation/simulation_backend.py:82-101 */
    const int n = orc_num_substeps(dt, dt_physics);
    const double dt_sub = dt / n;
    for (int i = 0; i < n; ++i) if (orc_rk4_step(P, x, c, dt_sub) != 0) return -1;
    return n;
}

void orc_derived(const double x[FD_NX], double d[FD_ND])
{   /* simulation/simplified_6dof.py:295-331 and _body_to_ned :505-530 */
    const double u = x[3], v = x[4], w = x[5], phi = x[6], theta = x[7], psi = x[8];
    d[FD_D_AIRSPEED] = sqrt(u * u + v * v + w * w);
    d[FD_D_ALTITUDE] = -x[2];
    const double r00 = cos(theta) * cos(psi);
    const double r01 = sin(phi) * sin(theta) * cos(psi) - cos(phi) * sin(psi);
    const double r02 = cos(phi) * sin(theta) * cos(psi) + sin(phi) * sin(psi);
    const double r10 = cos(theta) * sin(psi);
    const double r11 = sin(phi) * sin(theta) * sin(psi) + cos(phi) * cos(psi);
    const double r12 = cos(phi) * sin(theta) * sin(psi) - sin(phi) * cos(psi);
    const double vn = r00 * u + r01 * v + r02 * w;
    const double ve = r10 * u + r11 * v + r12 * w;
    d[FD_D_HEADING] = atan2(ve, vn);
    d[FD_D_GROUND_SPEED] = sqrt(vn * vn + ve * ve);
}

void orc_clip_controls(const double in[FD_NU], double out[FD_NU])
{   /* simulation/simplified_6dof.py:221-226 */
    out[FD_U_ELEVATOR] = clipd(in[FD_U_ELEVATOR], -1.0, 1.0);
    out[FD_U_AILERON] = clipd(in[FD_U_AILERON], -1.0, 1.0);
    out[FD_U_RUDDER] = clipd(in[FD_U_RUDDER], -1.0, 1.0);
    out[FD_U_THROTTLE] = clipd(in[FD_U_THROTTLE], 0.0, 1.0);
}

/* ---------- fp32 PID ----------------------------------------------------------------------------------- */
static inline float clampf(float v, float lo, float hi)
{   /* cpp/src/pid_controller.cpp:79-81  std::max(min_val, std::min(value, max_val)) */
    float m = (hi < v) ? hi : v;     /* std::min(value, max_val) */
    return (lo < m) ? m : lo;        /* std::max(min_val, m)     */
}

float orc_pid_compute(const float cfg[FD_NPC], float s[FD_NPS], float setpoint, float measurement, float dt)
{   /* cpp/src/pid_controller.cpp:24-60 */
    const float error = setpoint - measurement;
    const float p_term = cfg[FD_PC_KP] * error;
    float integral = s[FD_PS_INTEGRAL] + error * dt;
    integral = clampf(integral, cfg[FD_PC_INT_MIN], cfg[FD_PC_INT_MAX]);
    const float i_term = cfg[FD_PC_KI] * integral;
    float derivative;
    if (dt > 1e-6f) derivative = (error - s[FD_PS_ERR_PREV]) / dt; else derivative = 0.0f;
    const float a = cfg[FD_PC_ALPHA];
    const float dfilt = a * derivative + (1.0f - a) * s[FD_PS_DFILT];
    const float d_term = cfg[FD_PC_KD] * dfilt;
    float out = p_term + i_term + d_term;
    out = clampf(out, cfg[FD_PC_OUT_MIN], cfg[FD_PC_OUT_MAX]);
    s[FD_PS_INTEGRAL] = integral;
    s[FD_PS_ERR_PREV] = error;
    s[FD_PS_DFILT] = dfilt;
    return out;
}

#define PCFG(i) (pid_cfg + (i) * FD_NPC)
#define PST(i) (pid_state + (i) * FD_NPS)

/* ---------- cascade agents ----------------------------------------------------------------------------- */
void orc_rate_agent(const float *pid_cfg, float *pid_state, const double *C, const double rate_cmd[3],
                    double throttle, const double x[FD_NX], double dt, double surf[FD_NU])
{   /* controllers/rate_agent.py:92-122 */
    const double p_cmd = clipd(rate_cmd[0], -C[FD_C_MAX_ROLL_RATE], C[FD_C_MAX_ROLL_RATE]);
    const double q_cmd = clipd(rate_cmd[1], -C[FD_C_MAX_PITCH_RATE], C[FD_C_MAX_PITCH_RATE]);
    const double r_cmd = clipd(rate_cmd[2], -C[FD_C_MAX_YAW_RATE], C[FD_C_MAX_YAW_RATE]);
    const float fdt = (float)dt;   /* pybind narrows Python doubles to float */
    const float o_roll = orc_pid_compute(PCFG(FD_PID_RATE_ROLL), PST(FD_PID_RATE_ROLL), (float)p_cmd, (float)x[9], fdt);
    const float o_pitch = orc_pid_compute(PCFG(FD_PID_RATE_PITCH), PST(FD_PID_RATE_PITCH), (float)q_cmd, (float)x[10], fdt);
    const float o_yaw = orc_pid_compute(PCFG(FD_PID_RATE_YAW), PST(FD_PID_RATE_YAW), (float)r_cmd, (float)x[11], fdt);
    surf[FD_U_AILERON] = clipd((double)o_roll, -1.0, 1.0);
    surf[FD_U_ELEVATOR] = clipd(-(double)o_pitch, -1.0, 1.0);
    surf[FD_U_RUDDER] = clipd(-(double)o_yaw, -1.0, 1.0);
    surf[FD_U_THROTTLE] = clipd(throttle, 0.0, 1.0);
}

void orc_attitude_agent(const float *pid_cfg, float *pid_state, const double *C, const double ang_cmd[3],
                        int has_yaw, double throttle, const double x[FD_NX], double dt, double surf[FD_NU])
{   /* controllers/attitude_agent.py:116-152 */
    const double roll_cmd = clipd(ang_cmd[0], -C[FD_C_MAX_ROLL], C[FD_C_MAX_ROLL]);
    const double pitch_cmd = clipd(ang_cmd[1], -C[FD_C_MAX_PITCH], C[FD_C_MAX_PITCH]);
    const double yaw_cmd = has_yaw ? orc_wrap_angle(ang_cmd[2]) : 0.0;
    const double cur_yaw = orc_wrap_angle(x[8]);
    const float fdt = (float)dt;
    const float o_r = orc_pid_compute(PCFG(FD_PID_ATT_ROLL), PST(FD_PID_ATT_ROLL), (float)roll_cmd, (float)x[6], fdt);
    const float o_p = orc_pid_compute(PCFG(FD_PID_ATT_PITCH), PST(FD_PID_ATT_PITCH), (float)pitch_cmd, (float)x[7], fdt);
    const float o_y = orc_pid_compute(PCFG(FD_PID_ATT_YAW), PST(FD_PID_ATT_YAW), (float)yaw_cmd, (float)cur_yaw, fdt);
    double rate_cmd[3];
    rate_cmd[0] = clipd((double)o_r, -C[FD_C_MAX_ROLL_RATE], C[FD_C_MAX_ROLL_RATE]);
    rate_cmd[1] = clipd((double)o_p, -C[FD_C_MAX_PITCH_RATE], C[FD_C_MAX_PITCH_RATE]);
    rate_cmd[2] = clipd((double)o_y, -C[FD_C_MAX_YAW_RATE], C[FD_C_MAX_YAW_RATE]);
    orc_rate_agent(pid_cfg, pid_state, C, rate_cmd, throttle, x, dt, surf);
}

void orc_hsa_agent(const float *pid_cfg, float *pid_state, const double *C, const double hsa_cmd[3],
                   const double x[FD_NX], const double d[FD_ND], double dt, double surf[FD_NU])
{   /* controllers/hsa_agent.py:149-229 */
    const double heading = d[FD_D_HEADING];
    const double heading_error = orc_wrap_angle(hsa_cmd[0] - heading);
    const double virtual_setpoint = heading + heading_error;
    const float fdt = (float)dt;
    double roll_angle = (double)orc_pid_compute(PCFG(FD_PID_HEADING), PST(FD_PID_HEADING),
                                                (float)virtual_setpoint, (float)heading, fdt);
    roll_angle = clipd(roll_angle, -C[FD_C_MAX_BANK_RAD], C[FD_C_MAX_BANK_RAD]);

    const double g = 9.81, h = d[FD_D_ALTITUDE], V = d[FD_D_AIRSPEED];         /* :173-185 */
    const double h_cmd = hsa_cmd[2], V_cmd = hsa_cmd[1];
    const double E_specific = g * h + 0.5 * (V * V);
    const double E_specific_cmd = g * h_cmd + 0.5 * (V_cmd * V_cmd);
    const double E_balance = g * h - 0.5 * (V * V);
    const double E_balance_cmd = g * h_cmd - 0.5 * (V_cmd * V_cmd);

    const double thr_adj = (double)orc_pid_compute(PCFG(FD_PID_ENERGY), PST(FD_PID_ENERGY),
                                                   (float)E_specific_cmd, (float)E_specific, fdt);
    double throttle = C[FD_C_BASELINE_THROTTLE] + thr_adj;
    throttle = clipd(throttle, 0.0, 1.0);

    double pitch_angle = (double)orc_pid_compute(PCFG(FD_PID_BALANCE), PST(FD_PID_BALANCE),
                                                 (float)E_balance_cmd, (float)E_balance, fdt);
    const double cos_roll = cos(roll_angle);                                   /* :205-208 */
    if (fabs(cos_roll) > 0.01) {
        const double load_factor = 1.0 / cos_roll;
        pitch_angle += C[FD_C_LOAD_FACTOR_GAIN] * (load_factor - 1.0);
    }
    pitch_angle = clipd(pitch_angle, -C[FD_C_MAX_PITCH_CMD_RAD], C[FD_C_MAX_PITCH_CMD_RAD]);
    const double ang_cmd[3] = { roll_angle, pitch_angle, x[8] };               /* yaw_angle = state.yaw */
    orc_attitude_agent(pid_cfg, pid_state, C, ang_cmd, 1, throttle, x, dt, surf);
}

void orc_waypoint_agent(const float *pid_cfg, float *pid_state, const double *C, const double wp[FD_NWP],
                        const double x[FD_NX], const double d[FD_ND], double dt, double surf[FD_NU])
{   /* controllers/waypoint_agent.py:107-242 */
    const double e0 = wp[FD_WP_NORTH] - x[0], e1 = wp[FD_WP_EAST] - x[1];
    const double horizontal_distance = sqrt(e0 * e0 + e1 * e1);
    const double heading = d[FD_D_HEADING], airspeed = d[FD_D_AIRSPEED];
    const int gtype = (int)C[FD_C_GUIDANCE_TYPE];
    double heading_cmd;
    if (gtype == FD_GUIDANCE_LOS) {                                            /* :118-144 */
        heading_cmd = atan2(e1, e0);
        const double V = pymax(airspeed, 10.0);
        const double max_bank = C[FD_C_LOS_MAX_BANK_RAD];
        const double turn_radius = (V * V) / (9.81 * tan(max_bank));
        double heading_error = heading_cmd - heading;
        heading_error = atan2(sin(heading_error), cos(heading_error));
        const double anticipation_dist = turn_radius * fabs(heading_error) / deg2rad(90.0);
        const double lead = C[FD_C_LOS_LEAD_ANGLE_RAD];
        if (horizontal_distance < anticipation_dist && fabs(heading_error) > deg2rad(20.0)) {
            const double blend = 1.0 - (horizontal_distance / anticipation_dist);
            heading_cmd = heading_cmd + blend * lead * signd(heading_error);
            heading_cmd = atan2(sin(heading_cmd), cos(heading_cmd));
        }
    } else if (gtype == FD_GUIDANCE_PP) {                                      /* :146-178 */
        const double V = pymax(airspeed, 10.0);
        const double max_bank = C[FD_C_WP_MAX_BANK_RAD];
        const double turn_radius = (V * V) / (9.81 * tan(max_bank));
        double lookahead = C[FD_C_LOOKAHEAD_TIME] * V;
        const double proximity = C[FD_C_PROXIMITY_SCALE] * turn_radius;
        if (horizontal_distance < proximity) {
            const double scale = 0.6 + 0.4 * (horizontal_distance / proximity);
            lookahead *= scale;
        }
        lookahead = clipd(lookahead, C[FD_C_LOOKAHEAD_MIN], C[FD_C_LOOKAHEAD_MAX]);
        if (horizontal_distance > 1.0) {
            const double dir0 = e0 / horizontal_distance, dir1 = e1 / horizontal_distance;
            const double eff = pymin(lookahead, horizontal_distance);
            heading_cmd = atan2(dir1 * eff, dir0 * eff);
        } else {
            heading_cmd = atan2(e1, e0);
        }
    } else {
        heading_cmd = atan2(e1, e0);
    }
    heading_cmd = atan2(sin(heading_cmd), cos(heading_cmd));                   /* :185 */

    const double altitude_cmd = wp[FD_WP_ALTITUDE];
    double speed_cmd = isnan(wp[FD_WP_SPEED]) ? airspeed : wp[FD_WP_SPEED];     /* :207-211 (None -> airspeed) */
    const double heading_to_wp = atan2(e1, e0);                                /* :214-228 */
    double he = fabs(heading_to_wp - heading);
    he = fabs(atan2(sin(he), cos(he)));
    const double td = C[FD_C_TURN_THRESHOLD_DIST], ta = C[FD_C_TURN_THRESHOLD_ANGLE_RAD];
    if (horizontal_distance < td && he > ta) {
        const double reduction = C[FD_C_MAX_SPEED_REDUCTION] * (1.0 - horizontal_distance / td);
        speed_cmd = speed_cmd * (1.0 - reduction);
        speed_cmd = pymax(speed_cmd, C[FD_C_MIN_SPEED]);
    }
    const double hsa_cmd[3] = { heading_cmd, speed_cmd, altitude_cmd };
    orc_hsa_agent(pid_cfg, pid_state, C, hsa_cmd, x, d, dt, surf);
}

int orc_mission_update(const double *C, const double *wps, int n_wp, int32_t *wp_idx, const double x[FD_NX])
{   /* controllers/mission_planner.py:128-184,186-206 */
    if (*wp_idx >= n_wp) return 0;
    const double *wp = wps + (*wp_idx) * FD_NWP;
    const double e0 = wp[FD_WP_NORTH] - x[0], e1 = wp[FD_WP_EAST] - x[1], e2 = (-wp[FD_WP_ALTITUDE]) - x[2];
    const double dist = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    if (dist < C[FD_C_ACCEPTANCE_RADIUS]) { *wp_idx += 1; return 1; }
    return 0;
}

int orc_cascade_step(const double *P, const float *pid_cfg, float *pid_state, const double *C,
                     const double *wps, int n_wp, int32_t *wp_idx, double x[FD_NX], double dt,
                     double surf_out[FD_NU], int *reached)
{   /* examples/03_waypoint_square_demo.py:148-209 */
    double d[FD_ND], surf[FD_NU], cc[FD_NU];
    orc_derived(x, d);
    const int r = orc_mission_update(C, wps, n_wp, wp_idx, x);
    if (reached) *reached = r;
    if (*wp_idx >= n_wp) {
        if ((int)C[FD_C_ON_COMPLETE] == 1) *wp_idx = 0; else return 1;
    }
    orc_waypoint_agent(pid_cfg, pid_state, C, wps + (*wp_idx) * FD_NWP, x, d, dt, surf);
    orc_clip_controls(surf, cc);
    if (surf_out) memcpy(surf_out, surf, sizeof(surf));
    orc_rk4_step(P, x, cc, dt);
    return 0;
}

/* ---------- rate-control env ---------------------------------------------------------------------------- */
double orc_tracking_reward(double *e, const double err[3], const double action[4], double airspeed,
                           double altitude, double roll, double pitch, double comps[5])
{   /* learned_controllers/envs/rewards.py:75-137, default weights :19-25 */
    const double tracking_error = (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]) / 3.0;
    const double r_tracking = -0.5 * tracking_error;
    const double d0 = action[0] - e[FD_E_PREV_AIL], d1 = action[1] - e[FD_E_PREV_ELEV], d2 = action[2] - e[FD_E_PREV_RUD];
    const double control_change = d0 * d0 + (d1 * d1 + d2 * d2);   /* np.sum: a[0] + pairwise(a[1:]) */
    const double r_smoothness = -0.01 * control_change;
    const double roll_stability = exp(-fabs(roll) / deg2rad(45.0));
    const double pitch_stability = exp(-fabs(pitch) / deg2rad(30.0));
    const double airspeed_stability = clipd((airspeed - 8.0) / 12.0, 0.0, 1.0);
    const double altitude_stability = clipd((altitude - 10.0) / 90.0, 0.0, 1.0);
    const double stability_score = (roll_stability + pitch_stability + airspeed_stability + altitude_stability) / 4.0;
    const double r_stability = 0.3 * stability_score;
    for (int i = 0; i < 3; ++i) {
        const double pe = e[FD_E_PERR_P + i];
        const int flip = (signd(err[i]) != signd(pe)) && (fabs(pe) > 0.01);
        e[FD_E_SIGN_P + i] = 0.9 * e[FD_E_SIGN_P + i] + (flip ? 1.0 : 0.0);
        e[FD_E_PERR_P + i] = err[i];
    }
    const double osc = e[FD_E_SIGN_P] + (e[FD_E_SIGN_Q] + e[FD_E_SIGN_R]);       /* np.sum, as above */
    const double r_oscillation = -0.1 * osc;
    const double r_survival = 1.0;
    if (comps) { comps[0] = r_tracking; comps[1] = r_smoothness; comps[2] = r_stability; comps[3] = r_oscillation; comps[4] = r_survival; }
    return r_tracking + r_smoothness + r_stability + r_oscillation + r_survival;
}

double orc_settle_bonus(double *e, const double err[3], const double cmd[3], double dt)
{   /* learned_controllers/envs/rewards.py:193-221, defaults :162-166 */
    int settled = 1;
    for (int i = 0; i < 3; ++i) settled &= fabs(err[i]) < pymax(fabs(cmd[i]) * 0.05, 0.05);
    if (settled) {
        e[FD_E_SETTLE_TIMER] += dt;
        if (e[FD_E_SETTLE_TIMER] >= 0.2) { e[FD_E_IS_SETTLED] = 1.0; return 2.0 * dt; }
    } else {
        e[FD_E_SETTLE_TIMER] = 0.0;
        e[FD_E_IS_SETTLED] = 0.0;
    }
    return 0.0;
}

static void env_obs(const double x[FD_NX], const double e[FD_NE], float obs[FD_OBS_DIM])
{   /* learned_controllers/envs/rate_env.py:374-408 */
    double d[FD_ND];
    orc_derived(x, d);
    obs[0] = (float)x[9]; obs[1] = (float)x[10]; obs[2] = (float)x[11];
    obs[3] = (float)e[FD_E_CMD_P]; obs[4] = (float)e[FD_E_CMD_Q]; obs[5] = (float)e[FD_E_CMD_R];
    obs[6] = (float)(e[FD_E_CMD_P] - x[9]); obs[7] = (float)(e[FD_E_CMD_Q] - x[10]); obs[8] = (float)(e[FD_E_CMD_R] - x[11]);
    obs[9] = (float)d[FD_D_AIRSPEED]; obs[10] = (float)d[FD_D_ALTITUDE];
    obs[11] = (float)x[6]; obs[12] = (float)x[7]; obs[13] = (float)x[8];
    obs[14] = (float)e[FD_E_PREV_AIL]; obs[15] = (float)e[FD_E_PREV_ELEV];
    obs[16] = (float)e[FD_E_PREV_RUD]; obs[17] = (float)e[FD_E_PREV_THR];
}

void orc_env_reset(const double *EC, double x[FD_NX], double e[FD_NE], int32_t ei[FD_NEI],
                   const double rec[FD_NR], float obs[FD_OBS_DIM])
{   /* learned_controllers/envs/rate_env.py:170-210 with _generate_new_command :302-340 */
    const int ct = (int)EC[FD_EC_CMD_TYPE];
    memset(x, 0, sizeof(double) * FD_NX);
    x[2] = -rec[FD_R_ALTITUDE];
    x[3] = rec[FD_R_AIRSPEED];
    x[6] = rec[FD_R_ROLL]; x[7] = rec[FD_R_PITCH]; x[8] = rec[FD_R_YAW];
    x[9] = rec[FD_R_P]; x[10] = rec[FD_R_Q]; x[11] = rec[FD_R_R];
    memset(e, 0, sizeof(double) * FD_NE);
    if (ct == FD_CMD_STEP) {
        e[FD_E_CMD_P] = rec[FD_R_CMD0]; e[FD_E_CMD_Q] = rec[FD_R_CMD1]; e[FD_E_CMD_R] = rec[FD_R_CMD2];
    } else if (ct == FD_CMD_RAMP || ct == FD_CMD_SINE) {
        e[FD_E_SCHED0] = rec[FD_R_CMD0]; e[FD_E_SCHED1] = rec[FD_R_CMD1]; e[FD_E_SCHED2] = rec[FD_R_CMD2];
        e[FD_E_SCHED3] = rec[FD_R_CMD3];
    }
    e[FD_E_PREV_THR] = 0.5;                                                    /* :193 */
    ei[FD_EI_STEP] = 0;
    env_obs(x, e, obs);
}

void orc_env_step(const double *P, const double *EC, double x[FD_NX], double e[FD_NE], int32_t ei[FD_NEI],
                  const float action[FD_ACT_DIM], const double rw_delta[3], float obs[FD_OBS_DIM],
                  double *reward, int32_t *terminated, int32_t *truncated)
{   /* learned_controllers/envs/rate_env.py:212-300 */
    const double dt = EC[FD_EC_DT];
    double a[4];
    a[0] = clipd((double)action[0], -1.0, 1.0); a[1] = clipd((double)action[1], -1.0, 1.0);   /* :225 */
    a[2] = clipd((double)action[2], -1.0, 1.0); a[3] = clipd((double)action[3], 0.0, 1.0);
    double surf[FD_NU], cc[FD_NU];
    surf[FD_U_AILERON] = a[0]; surf[FD_U_ELEVATOR] = a[1]; surf[FD_U_RUDDER] = a[2]; surf[FD_U_THROTTLE] = a[3];
    orc_clip_controls(surf, cc);                                               /* :234 */
    orc_backend_step(P, x, cc, dt, EC[FD_EC_DT_PHYSICS]);                      /* :237 */
    e[FD_E_TIME] += dt;                                                        /* :241-242 */
    ei[FD_EI_STEP] += 1;

    const int ct = (int)EC[FD_EC_CMD_TYPE];                                    /* _update_command :342-372 */
    if (ct == FD_CMD_RAMP) {
        const double t = e[FD_E_TIME] - 0.0, duration = 3.0;
        if (t < duration) {
            const double alpha = t / duration;
            for (int i = 0; i < 3; ++i) e[FD_E_CMD_P + i] = (1 - alpha) * 0.0 + alpha * e[FD_E_SCHED0 + i];
        } else {
            for (int i = 0; i < 3; ++i) e[FD_E_CMD_P + i] = e[FD_E_SCHED0 + i];
        }
    } else if (ct == FD_CMD_RANDOM_WALK) {
        for (int i = 0; i < 3; ++i) {
            const double m = EC[FD_EC_MAX_RATE_P + i];
            e[FD_E_CMD_P + i] = clipd(e[FD_E_CMD_P + i] + rw_delta[i], -m, m);
        }
    } else if (ct == FD_CMD_SINE) {
        const double s = sin(2 * M_PI * e[FD_E_SCHED3] * e[FD_E_TIME]);
        for (int i = 0; i < 3; ++i) e[FD_E_CMD_P + i] = e[FD_E_SCHED0 + i] * s;
    }

    double d[FD_ND];
    orc_derived(x, d);
    const double cmd[3] = { e[FD_E_CMD_P], e[FD_E_CMD_Q], e[FD_E_CMD_R] };
    const double err[3] = { cmd[0] - x[9], cmd[1] - x[10], cmd[2] - x[11] };
    double r = orc_tracking_reward(e, err, a, d[FD_D_AIRSPEED], d[FD_D_ALTITUDE], x[6], x[7], 0);
    r += orc_settle_bonus(e, err, cmd, dt);
    e[FD_E_PREV_AIL] = a[0]; e[FD_E_PREV_ELEV] = a[1]; e[FD_E_PREV_RUD] = a[2]; e[FD_E_PREV_THR] = a[3];   /* :282 */

    const int term = (d[FD_D_ALTITUDE] < 5.0) || (fabs(x[6]) > deg2rad(120.0)) ||          /* :437-460 */
                     (fabs(x[7]) > deg2rad(80.0)) || (d[FD_D_AIRSPEED] < 8.0);
    const int trunc = ei[FD_EI_STEP] >= (int)EC[FD_EC_MAX_STEPS];
    if (term && !trunc) r += -100.0;                                           /* :289-292 */
    e[FD_E_EP_RETURN] += r;
    env_obs(x, e, obs);
    *reward = r; *terminated = term; *truncated = trunc;
}

void orc_residual_env_step(const double *P, const double *EC, const float *pid_cfg, float *pid_state, const double *C,
                           double x[FD_NX], double e[FD_NE], int32_t ei[FD_NEI], const float residual[FD_ACT_DIM],
                           float scale, const double rw_delta[3], float obs[FD_OBS_DIM], double *reward,
                           int32_t *terminated, int32_t *truncated, float pid_action_out[FD_ACT_DIM])
{   /* learned_controllers/envs/residual_rate_env.py:99-157 (float32 action arithmetic as NumPy does it) */
    const double cmd[3] = { e[FD_E_CMD_P], e[FD_E_CMD_Q], e[FD_E_CMD_R] };
    double surf[FD_NU];
    orc_rate_agent(pid_cfg, pid_state, C, cmd, C[FD_C_PID_THROTTLE], x, C[FD_C_PID_DT] > 0.0 ? C[FD_C_PID_DT] : EC[FD_EC_DT], surf);
    const float pid_a[4] = { (float)surf[FD_U_AILERON], (float)surf[FD_U_ELEVATOR], (float)surf[FD_U_RUDDER], (float)surf[FD_U_THROTTLE] };
    float comb[4];
    for (int i = 0; i < 4; ++i) {
        const float c = pid_a[i] + residual[i] * scale;
        const float lo = i < 3 ? -1.0f : 0.0f;
        comb[i] = c < lo ? lo : (c > 1.0f ? 1.0f : c);
        if (pid_action_out) pid_action_out[i] = pid_a[i];
    }
    orc_env_step(P, EC, x, e, ei, comb, rw_delta, obs, reward, terminated, truncated);
    const float mag = residual[0] * residual[0] + (residual[1] * residual[1] + residual[2] * residual[2]);
    const double bonus = 0.05 * (1.0 - (double)mag / 3.0);   /* numpy<2: float32 scalar / python float -> float64 */
    *reward += bonus;
    e[FD_E_EP_RETURN] += bonus;
}

/* ---------- evaluation metrics: learned_controllers/eval/metrics.py:95-362 ---------------------------------- */
/* NumPy's pairwise summation of a strided fp64 vector (what np.sum / np.mean run for these reductions), restated so the
 * oracle is bit-comparable with the reference's outputs: < 8 elements sequential; <= 128 eight running partial sums
 * combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; longer inputs split in halves (multiple of 8). */
static double np_pairwise(const double *a, long n, long stride)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i * stride];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i * stride];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2, stride) + np_pairwise(a + n2 * stride, n - n2, stride);
}

void orc_rate_metrics(const double *times, const double *rates, const double *commands, const double *actions,
                      const double *rewards, int len, double settling_threshold, int settle_steps, double out[FD_NM])
{
    for (int k = 0; k < FD_NM; ++k) out[k] = 0.0;
    if (len <= 0) return;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)len * 3);
    const double t_end = times[len - 1];
    for (int ax = 0; ax < 3; ++ax) {                                              /* metrics.py:138-163 */
        double max_cmd = 0.0;
        for (int i = 0; i < len; ++i) max_cmd = fmax(max_cmd, fabs(commands[i * 3 + ax]));
        if (max_cmd < 0.01) continue;                                             /* :144-145 */
        /* _compute_settling_time :182-216 */
        const double thr = pymax(settling_threshold * max_cmd, 0.05);
        double settle = t_end;
        for (int i = 0; i < len - settle_steps; ++i) {
            int all = 1;
            for (int j = i; j < i + settle_steps; ++j)
                if (!(fabs(commands[j * 3 + ax] - rates[j * 3 + ax]) < thr)) { all = 0; break; }
            if (all) { settle = times[i]; break; }
        }
        out[FD_M_SETTLE_ROLL + ax] = settle;
        /* _compute_overshoot :218-257 */
        const double sgn = signd(commands[(len / 2) * 3 + ax]);
        double over = 0.0;
        if (sgn != 0.0) {
            double mx = 0.0; int any = 0;
            for (int i = 0; i < len; ++i) {
                const double err = rates[i * 3 + ax] - commands[i * 3 + ax];
                if (sgn * err > 0.0) { any = 1; mx = fmax(mx, fabs(err)); }
            }
            if (any) over = (mx / max_cmd) * 100.0;
        }
        out[FD_M_OVERSHOOT_ROLL + ax] = over;
        /* _compute_rise_time :259-294 */
        const int n_ss = len / 5 > 1 ? len / 5 : 1;
        const double cmd_ss = np_pairwise(commands + (size_t)(len - n_ss) * 3 + ax, n_ss, 3) / (double)n_ss;
        double rise = 0.0;
        if (!(fabs(cmd_ss) < 0.01)) {
            const double th = 0.9 * cmd_ss;
            rise = t_end;
            for (int i = 0; i < len; ++i) {
                const double r = rates[i * 3 + ax];
                if (cmd_ss > 0.0 ? r >= th : r <= th) { rise = times[i]; break; }
            }
        }
        out[FD_M_RISE_ROLL + ax] = rise;
        /* _compute_steady_state_error :296-321 ; np.searchsorted(times, settle) (left) */
        int idx = 0;
        while (idx < len && times[idx] < settle) ++idx;
        for (int i = 0; i < len; ++i) tmp[i] = fabs(commands[i * 3 + ax] - rates[i * 3 + ax]);
        if (idx >= len - 1) out[FD_M_SSERR_ROLL + ax] = np_pairwise(tmp, len, 1) / (double)len;
        else out[FD_M_SSERR_ROLL + ax] = np_pairwise(tmp + idx, len - idx, 1) / (double)(len - idx);
    }
    /* _compute_smoothness :323-340 (surfaces only); a one-step episode gives NumPy's mean of nothing = NaN */
    for (int i = 0; i + 1 < len; ++i)
        for (int k = 0; k < 3; ++k) tmp[i * 3 + k] = fabs(actions[(i + 1) * 4 + k] - actions[i * 4 + k]);
    out[FD_M_SMOOTHNESS] = len > 1 ? np_pairwise(tmp, (long)(len - 1) * 3, 1) / (double)((len - 1) * 3) : NAN;
    /* _compute_tracking_rmse :342-358 */
    for (int i = 0; i < len * 3; ++i) { const double e = commands[i] - rates[i]; tmp[i] = e * e; }
    out[FD_M_RMSE] = sqrt(np_pairwise(tmp, (long)len * 3, 1) / (double)(len * 3));
    out[FD_M_SUCCESS] = (out[FD_M_SETTLE_ROLL] < t_end && out[FD_M_SETTLE_PITCH] < t_end &&
                         out[FD_M_SETTLE_YAW] < t_end) ? 1.0 : 0.0;               /* :171-176 */
    out[FD_M_EPISODE_LENGTH] = t_end;
    out[FD_M_TOTAL_REWARD] = np_pairwise(rewards, len, 1);
    free(tmp);
}

/* ---------- sensor layer: interfaces/sensor.py:199-243 (NoisySensorInterface.update) ------------------------------ */
void orc_sensor_update(const double x[FD_NX], double airspeed, double altitude, double bias[FD_NSB],
                       const double cfg[FD_NSN], const double z[FD_NSZ], double meas[FD_NMS])
{   /* rng.normal(0, s, k) is 0 + s * standard_normal: z holds the standard normals in the reference's call order.
     * airspeed / altitude: the true state's derived fields (AircraftState.airspeed, .altitude) */
    if (cfg[FD_SN_ENABLED] == 0.0) {                                              /* :203-205 */
        for (int k = 0; k < FD_NX; ++k) meas[k] = x[k];
        meas[FD_MS_AIRSPEED] = airspeed; meas[FD_MS_ALTITUDE] = altitude;
        return;
    }
    for (int k = 0; k < 3; ++k) {
        meas[FD_X_N + k] = x[FD_X_N + k] + (0.0 + cfg[FD_SN_GPS_POS] * z[FD_SZ_POS + k]);            /* :208-210 */
        meas[FD_X_U + k] = x[FD_X_U + k] + (0.0 + cfg[FD_SN_GPS_VEL] * z[FD_SZ_VEL + k]);            /* :211-213 */
        meas[FD_X_ROLL + k] = x[FD_X_ROLL + k] + (0.0 + cfg[FD_SN_ATTITUDE] * z[FD_SZ_ATT + k]);     /* :214-216 */
        meas[FD_X_P + k] = (x[FD_X_P + k] + (0.0 + cfg[FD_SN_GYRO] * z[FD_SZ_GYRO + k])) + bias[k];  /* :217-221 */
    }
    meas[FD_MS_AIRSPEED] = airspeed + (0.0 + cfg[FD_SN_AIRSPEED] * z[FD_SZ_AIRSPEED]);               /* :222-224 */
    meas[FD_MS_ALTITUDE] = altitude + (0.0 + cfg[FD_SN_ALTITUDE] * z[FD_SZ_ALTITUDE]);               /* :225-227 */
    for (int k = 0; k < 3; ++k) {                                                                     /* :230-231 */
        bias[k] += 0.0 + cfg[FD_SN_GYRO_BIAS_WALK] * z[FD_SZ_GYRO_BIAS + k];
        bias[3 + k] += 0.0 + cfg[FD_SN_ACCEL_BIAS_WALK] * z[FD_SZ_ACCEL_BIAS + k];
    }
}

/* ---------- batch drivers (SoA) -------------------------------------------------------------------------- */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_sixdof_step_batch(const double *P, double *xs, const double *us, int64_t n, double dt, int n_sub,
                           int n_threads)
{
    const double dt_sub = dt / n_sub;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double x[12], u[4], c[4];
        for (int k = 0; k < 12; ++k) x[k] = xs[k * n + i];
        for (int k = 0; k < 4; ++k) u[k] = us[k * n + i];
        orc_clip_controls(u, c);
        for (int s = 0; s < n_sub; ++s) orc_rk4_step(P, x, c, dt_sub);
        for (int k = 0; k < 12; ++k) xs[k * n + i] = x[k];
    }
}

void orc_env_step_batch(const double *P, const double *EC, double *xs, double *es, int32_t *eis,
                        const float *actions, float *obs, double *rewards, int32_t *terminated,
                        int32_t *truncated, int64_t n, int n_threads)
{
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double x[12], e[FD_NE], rw[3] = { 0, 0, 0 };
        int32_t ei[FD_NEI];
        for (int k = 0; k < 12; ++k) x[k] = xs[k * n + i];
        for (int k = 0; k < FD_NE; ++k) e[k] = es[k * n + i];
        for (int k = 0; k < FD_NEI; ++k) ei[k] = eis[k * n + i];
        orc_env_step(P, EC, x, e, ei, actions + i * 4, rw, obs + i * FD_OBS_DIM, rewards + i, terminated + i,
                     truncated + i);
        for (int k = 0; k < 12; ++k) xs[k * n + i] = x[k];
        for (int k = 0; k < FD_NE; ++k) es[k * n + i] = e[k];
        for (int k = 0; k < FD_NEI; ++k) eis[k * n + i] = ei[k];
    }
}

void orc_cascade_step_batch(const double *P, const float *pid_cfg, float *pss, const double *C,
                            const double *wps, int n_wp, int32_t *wp_idx, double *xs, double *surfs,
                            int64_t n, double dt, int n_steps, int n_threads)
{
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double x[12], surf[4] = { 0, 0, 0, 0 };
        float ps[FD_NPID * FD_NPS];
        int32_t idx = wp_idx[i];
        for (int k = 0; k < 12; ++k) x[k] = xs[k * n + i];
        for (int k = 0; k < FD_NPID * FD_NPS; ++k) ps[k] = pss[k * n + i];
        for (int s = 0; s < n_steps; ++s)
            if (orc_cascade_step(P, pid_cfg, ps, C, wps, n_wp, &idx, x, dt, surf, 0)) break;
        for (int k = 0; k < 12; ++k) xs[k * n + i] = x[k];
        for (int k = 0; k < FD_NPID * FD_NPS; ++k) pss[k * n + i] = ps[k];
        if (surfs) for (int k = 0; k < 4; ++k) surfs[k * n + i] = surf[k];
        wp_idx[i] = idx;
    }
}
