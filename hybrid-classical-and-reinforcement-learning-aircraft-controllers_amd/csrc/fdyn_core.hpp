// fdyn_core.hpp -- per-aircraft device code for gfx950 (MI355X): 6-DOF dynamics + RK4, fp32 PID, the
// 5-level cascade and the rate-control env step.  One lane == one aircraft; everything lives in registers.
//
// Precision is a template pair <S, T>:
//   S  = type the 12-word state is STORED and ACCUMULATED in (and the fp64/fp32 "glue" of agents / rewards)
//   T  = type one dynamics evaluation is COMPUTED in
//   <double,double> "f64"   : the parity variant (the reference is float64, simplified_6dof.py:173)
//   <double,float>  "mixed" : fp32 derivative evaluations, fp64 state accumulate (drift ~1e-6)
//   <float,float>   "f32"   : pure fp32 throughput variant
// The PID is always fp32 with contraction off: bit-faithful to cpp/src/pid_controller.cpp:24-60.
//
// Reference citations are relative to the reference root; layouts in include/fdyn_layout.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fdyn_layout.h"

namespace fdyn {

#define FD_DEV __device__ __forceinline__

// ----- scalar math traits ---------------------------------------------------------------------------------
template <typename T> struct M;
template <> struct M<double> {
    static FD_DEV double sin(double x) { return ::sin(x); }
    static FD_DEV double cos(double x) { return ::cos(x); }
    static FD_DEV void sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
    static FD_DEV double tan(double x) { return ::tan(x); }
    static FD_DEV double atan2(double y, double x) { return ::atan2(y, x); }
    static FD_DEV double asin(double x) { return ::asin(x); }
    static FD_DEV double sqrt(double x) { return ::sqrt(x); }
    static FD_DEV double exp(double x) { return ::exp(x); }
    static FD_DEV double abs(double x) { return ::fabs(x); }
    static FD_DEV double copysign(double a, double y) { return __builtin_copysign(a, y); }
    static FD_DEV double rint(double x) { return ::rint(x); }
    static FD_DEV double fmod(double x, double y) { return ::fmod(x, y); }
    static FD_DEV bool finite(double x) { return ::isfinite(x); }
};
// fp32: hand-rolled branch-free kernels.  ocml's sinf/cosf/atan2f inline the Payne-Hanek large-argument path and
// evaluate it under selects on every call (>1000 VALU per dynamics evaluation, measured); flight angles are bounded
// (|x| < ~10 rad before the per-step re-wrap), so a two-constant Cody-Waite reduction is exact enough.  Errors are
// ~1 ulp (<= 1.2e-7 abs on sin/cos), i.e. the same order as the fp32 rounding the variant already accepts.
namespace fast {
FD_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }          // v_rcp_f32, 1 ulp
// max of two values that are numbers: ONE v_max_f32.  __builtin_fmaxf is llvm.maxnum, which in IEEE mode first canonicalises
// every operand the compiler cannot prove quiet (v_max_f32 x, x -- two wasted instructions per call on a sqrt result and a
// staged parameter), and at one wave per SIMD an instruction is 5 cycles whatever it does.
// (v_med3_f32 with +inf as the third operand; NOT inline assembly: the compiler inserts the wait state a VALU instruction needs
// after the transcendental that produced its operand only for instructions it can see -- a v_max in an asm block right after
// v_sqrt read the old register contents.)
FD_DEV float max_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
// keep a value out of the compiler's if-conversion heuristics: computed here, unconditionally (a select whose expensive arm
// every lane needs anyway must not become a divergent branch: saveexec / xor / andn2 / cbranch + hazard nops per use)
FD_DEV float pinned(float x) { asm volatile("" : "+v"(x)); return x; }
FD_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }        // v_sqrt_f32, 1 ulp
FD_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }          // v_rsq_f32, 1 ulp
FD_DEV void sincos(float x, float& s, float& c)
{
    const float k = __builtin_rintf(x * 0.63661977236758134f);          // nearest multiple of pi/2
    float r = __builtin_fmaf(k, -1.5707962512969971f, x);               // pi/2 split hi + lo (Cody-Waite)
    r = __builtin_fmaf(k, -7.5497894158615964e-08f, r);
    const float z = r * r;
    // minimax on [-pi/4, pi/4] (Cephes sinf/cosf coefficients)
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(z * r, ps, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    const int q = int(k);
    const bool swap = q & 1;
    const float ss = swap ? cr : sr, cc = swap ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
// rotate (s, c) = (sin a, cos a) by a SMALL angle d (|d| <= 0.125): sin/cos of d from their Taylor series (truncation
// d^7/5040 < 1e-10, d^6/720 < 6e-9 -- below the fp32 ulp of the result), 10 VALU instead of a 25-instruction sincos.
// The RK4 stage states differ from the step's initial state by (dt/2 or dt) * euler rate, and consecutive sub-steps by
// dt/6 * (k1 + 2 k2 + 2 k3 + k4): all the trigonometry of a sub-step comes from ONE sincos per angle per launch.
// The two polynomial chains run as the halves of v_pk_fma_f32 (gfx950 packed fp32: two FMAs per instruction, and at one wave
// per SIMD an instruction is ~5 cycles whatever it does): 4 + 2 instructions per rotation instead of 6 + 2; same values.
typedef float f32x2 __attribute__((ext_vector_type(2)));
FD_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
FD_DEV f32x2 bc(float x) { return (f32x2){ x, x }; }
// (x, undefined): an operand of which the packed instruction reads the LOW half for both results -- no copy to build (x, x)
// r.y deliberately unset -- the packed instruction never reads it (op_sel_hi 0 for that operand).  (x, x) costs a v_mov per use,
// and so does a frozen unspecified value (__builtin_nondeterministic_value: +8 instructions per sub-step).
FD_DEV f32x2 lo_only(float x) { f32x2 r; r.x = x; return r; }
// A plane rotation negates ONE half of a pair.  The hardware has the per-half modifier (neg_lo / neg_hi); LLVM does not select
// it for packed fp32 (it emits v_pk_add_f32 x, 0 neg + v_mov: two instructions per negated half), so the two forms a rotation
// needs are spelled as ONE instruction each.  Their operands come from plain VALU instructions only (FMAs and products), never
// straight from a transcendental: the compiler does not see into the assembly to insert that wait state.
//   a * b.xx + (c.x, -c.y)
FD_DEV f32x2 pk_fma_bx_nhi(f32x2 a, f32x2 b, f32x2 c)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
//   a.yx * b.xx + (-c.x, c.y)
FD_DEV f32x2 pk_fma_ayx_bx_nlo(f32x2 a, f32x2 b, f32x2 c)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// rotate sc = (sin a, cos a) by a SMALL angle d: (s cd + c sd, c cd - s sd), the products and FMAs of the scalar form
// (so = fma(s, cd, c sd), co = fma(c, cd, -(s sd))) as one packed product + one packed FMA: 6 instructions per rotation
FD_DEV f32x2 rotate_small(f32x2 sc, float d)
{
    const float z = d * d;
    const f32x2 p1 = pk_fma(bc(z), (f32x2){ 4.1666667908e-2f, 8.3333337680e-3f }, (f32x2){ -0.5f, -1.6666667163e-1f });
    const f32x2 p2 = pk_fma(bc(z), p1, (f32x2){ 1.0f, 0.0f });       // (cos d, (sin d - d) / d)
    const float sd = __builtin_fmaf(p2.y, d, d);
    return pk_fma_bx_nhi(sc, p2, sc.yx * bc(sd));
}
// cos(x): Taylor to x^8 while |x| <= 0.8 rad (truncation < 3e-8; a bank command is limited to 25 deg), full reduction beyond
FD_DEV float cos_bounded(float x)
{
    const float z = x * x;
    float p = __builtin_fmaf(z, 2.4801587e-5f, -1.3888889e-3f);
    p = __builtin_fmaf(z, p, 4.1666668e-2f);
    p = __builtin_fmaf(z, p, -0.5f);
    float c = __builtin_fmaf(z, p, 1.0f);
    if (__builtin_expect(!(z <= 0.64f), 0)) { float sn; sincos(x, sn, c); }
    return c;
}
// atan(t) for |t| <= 0.7 with NO range reduction: t + t z P(z), z = t^2, degree-5 weighted least-squares fit on Chebyshev
// nodes (max abs error 4.7e-8 in fp32 evaluation; scripts in DESIGN.md §4).  0.7 = tan(35 deg) covers the angle-of-attack
// clip of every shipped aircraft type (30 deg), so alpha never needs the three-range atan2 in flight.
#define FD_ATAN_WIDE_LIMIT 0.7f
FD_DEV float atan_wide(float t)
{
    const float z = t * t;
    float p = __builtin_fmaf(z, 2.0541535690e-02f, -6.1546623707e-02f);
    p = __builtin_fmaf(z, p, 1.0221967846e-01f);
    p = __builtin_fmaf(z, p, -1.4136675000e-01f);
    p = __builtin_fmaf(z, p, 1.9987617433e-01f);
    p = __builtin_fmaf(z, p, -3.3332955837e-01f);
    return __builtin_fmaf(p * z, t, t);
}
// asin(x) for |x| <= 0.75: x + x z P(z), degree-7 fit of the same kind (max abs error 5.1e-8)
#define FD_ASIN_WIDE_LIMIT 0.75f
FD_DEV float asin_wide(float x)
{
    const float z = x * x;
    float p = __builtin_fmaf(z, 2.1202674508e-01f, -3.1944271922e-01f);
    p = __builtin_fmaf(z, p, 2.6334178448e-01f);
    p = __builtin_fmaf(z, p, -7.6388612390e-02f);
    p = __builtin_fmaf(z, p, 5.3024884313e-02f);
    p = __builtin_fmaf(z, p, 4.1755288839e-02f);
    p = __builtin_fmaf(z, p, 7.5183674693e-02f);
    p = __builtin_fmaf(z, p, 1.6666238010e-01f);
    return __builtin_fmaf(p * z, x, x);
}
// the same polynomial on (t, z = t^2) given separately (the caller's z may be cheaper or more accurate than t * t)
FD_DEV float asin_wide_t(float t, float z)
{
    float p = __builtin_fmaf(z, 2.1202674508e-01f, -3.1944271922e-01f);
    p = __builtin_fmaf(z, p, 2.6334178448e-01f);
    p = __builtin_fmaf(z, p, -7.6388612390e-02f);
    p = __builtin_fmaf(z, p, 5.3024884313e-02f);
    p = __builtin_fmaf(z, p, 4.1755288839e-02f);
    p = __builtin_fmaf(z, p, 7.5183674693e-02f);
    p = __builtin_fmaf(z, p, 1.6666238010e-01f);
    return __builtin_fmaf(p * z, t, t);
}
// atan_wide(ta) and asin_wide_t(tb, zb) in one go: (atan(ta), asin-polynomial(tb, zb)); every FMA is the one the two scalar
// functions make, the last four Horner steps, the p z product and the closing FMA two per v_pk_* instruction
FD_DEV f32x2 atan_asin_wide(float ta, float za, float tb, float zb)
{
    const float pa = __builtin_fmaf(za, 2.0541535690e-02f, -6.1546623707e-02f);
    float pb = __builtin_fmaf(zb, 2.1202674508e-01f, -3.1944271922e-01f);
    pb = __builtin_fmaf(zb, pb, 2.6334178448e-01f);
    pb = __builtin_fmaf(zb, pb, -7.6388612390e-02f);
    const f32x2 z = { za, zb }, t = { ta, tb };
    f32x2 p = pk_fma(z, (f32x2){ pa, pb }, (f32x2){ 1.0221967846e-01f, 5.3024884313e-02f });
    p = pk_fma(z, p, (f32x2){ -1.4136675000e-01f, 4.1755288839e-02f });
    p = pk_fma(z, p, (f32x2){ 1.9987617433e-01f, 7.5183674693e-02f });
    p = pk_fma(z, p, (f32x2){ -3.3332955837e-01f, 1.6666238010e-01f });
    return pk_fma(p * z, t, t);
}
FD_DEV float atan_pos(float a)
{   // atan for a >= 0 (Cephes atanf: two range reductions + degree-4 polynomial in a^2), branch-free
    const bool big = a > 2.414213562373095f, mid = a > 0.4142135623730950f;
    const float num = big ? -1.0f : (mid ? a - 1.0f : a);
    const float den = big ? a : (mid ? a + 1.0f : 1.0f);
    const float base = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
    const float t = num * rcp(den);
    const float z = t * t;
    float p = __builtin_fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = __builtin_fmaf(z, p, 1.99777106478e-1f);
    p = __builtin_fmaf(z, p, -3.33329491539e-1f);
    return base + __builtin_fmaf(p * z, t, t);
}
FD_DEV float atan2(float y, float x)
{
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    float a = atan_pos(mn * rcp(mx));                                   // in [0, pi/4]; NaN if both are 0
    a = (mx == 0.0f) ? 0.0f : a;
    a = (ay > ax) ? 1.5707963267948966f - a : a;
    a = (x < 0.0f) ? 3.14159265358979323846f - a : a;
    return __builtin_copysignf(a, y);
}
FD_DEV float asin_poly(float zz)
{   // (asin(t) - t) / t^3 on t^2 = zz <= 0.25 (Cephes asinf coefficients)
    float p = __builtin_fmaf(zz, 4.2163199048e-2f, 2.4181311049e-2f);
    p = __builtin_fmaf(zz, p, 4.5470025998e-2f);
    p = __builtin_fmaf(zz, p, 7.4953002686e-2f);
    return __builtin_fmaf(zz, p, 1.6666752422e-1f);
}
FD_DEV float asin_small(float x)                    // |x| <= 0.5 (beyond it the caller takes asin_from_one_minus)
{
    const float zz = x * x;
    return __builtin_fmaf(asin_poly(zz) * zz, x, x);
}
FD_DEV float asin_from_one_minus(float om)          // asin(1 - om) for om in [0, 0.5]: pi/2 - 2 asin(sqrt(om / 2))
{
    const float zz = 0.5f * om;
    const float t = sqrt(zz);
    const float r = __builtin_fmaf(asin_poly(zz) * zz, t, t);
    return __builtin_fmaf(-2.0f, r, 1.5707963267948966f);
}
FD_DEV float asin(float x)
{   // Cephes asinf: |x| > 0.5 -> pi/2 - 2 asin(sqrt((1-|x|)/2))
    const float ax = __builtin_fabsf(x);
    const bool big = ax > 0.5f;
    const float zz = big ? 0.5f * (1.0f - ax) : ax * ax;
    const float t = big ? sqrt(zz) : ax;
    float p = __builtin_fmaf(zz, 4.2163199048e-2f, 2.4181311049e-2f);
    p = __builtin_fmaf(zz, p, 4.5470025998e-2f);
    p = __builtin_fmaf(zz, p, 7.4953002686e-2f);
    p = __builtin_fmaf(zz, p, 1.6666752422e-1f);
    float r = __builtin_fmaf(p * zz, t, t);
    r = big ? 1.5707963267948966f - 2.0f * r : r;
    return __builtin_copysignf(r, x);
}
}  // namespace fast

template <> struct M<float> {
    static FD_DEV float sin(float x) { float s, c; fast::sincos(x, s, c); return s; }
    static FD_DEV float cos(float x) { float s, c; fast::sincos(x, s, c); return c; }
    static FD_DEV void sincos(float x, float& s, float& c) { fast::sincos(x, s, c); }
    static FD_DEV float tan(float x) { float s, c; fast::sincos(x, s, c); return s * fast::rcp(c); }
    static FD_DEV float atan2(float y, float x) { return fast::atan2(y, x); }
    static FD_DEV float asin(float x) { return fast::asin(x); }
    static FD_DEV float sqrt(float x) { return fast::sqrt(x); }
    static FD_DEV float exp(float x) { return __expf(x); }
    static FD_DEV float abs(float x) { return __builtin_fabsf(x); }
    static FD_DEV float copysign(float a, float y) { return __builtin_copysignf(a, y); }
    static FD_DEV float rint(float x) { return __builtin_rintf(x); }
    static FD_DEV float fmod(float x, float y) { return ::fmodf(x, y); }
    static FD_DEV bool finite(float x) { return __builtin_isfinite(x); }
};

// division: IEEE-correct for fp64 (parity), one v_rcp_f32 + multiply for fp32
FD_DEV double fdiv(double a, double b) { return a / b; }
FD_DEV float fdiv(float a, float b) { return a * fast::rcp(b); }

// Python / NumPy semantics the reference relies on
template <typename T> FD_DEV T clipv(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }   // np.clip, NaN stays
template <typename T> FD_DEV T pymax(T a, T b) { return b > a ? b : a; }                            // max(a, b)
template <typename T> FD_DEV T pymin(T a, T b) { return b < a ? b : a; }                            // min(a, b)
// single-instruction forms for the fp32 evaluation path (v_med3_f32 / v_max_f32).  They do not propagate NaN the way
// np.clip does, which is why dynamics<float> tests finiteness BEFORE its derivative clamps.
FD_DEV float clipf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
FD_DEV double clipf(double x, double lo, double hi) { return clipv(x, lo, hi); }
FD_DEV float maxf(float a, float b) { return __builtin_fmaxf(a, b); }
FD_DEV double maxf(double a, double b) { return pymax(a, b); }
template <typename T> FD_DEV T signv(T x) { return x > T(0) ? T(1) : (x < T(0) ? T(-1) : x); }      // np.sign (0->0, NaN->NaN)

#define FD_PI 3.14159265358979323846
template <typename T> FD_DEV T deg2rad(double d) { return T(d * (FD_PI / 180.0)); }   // constant-folded on the host side

// (a + pi) % (2 pi) - pi with Python floor-mod: [-pi, pi)     controllers/utils/pid_utils.py:65
template <typename T> FD_DEV T wrap_angle(T a)
{
    if constexpr (sizeof(T) == 4) {
        // fp32 glue of the reduced-precision variants: range reduction by rint (differs from the floor-mod form only in
        // which of -pi / +pi represents the half-turn itself); ocml's fmodf is ~40 instructions
        return __builtin_fmaf(-6.2831853071795865f, __builtin_rintf(a * 0.15915494309189535f), a);
    } else {
        const T b = T(2.0 * FD_PI);
        T m = M<T>::fmod(a + T(FD_PI), b);
        if (m < T(0)) m += b;
        return m - T(FD_PI);
    }
}

// ----- aircraft parameter block, converted to the compute type and held in registers ---------------------
template <typename T> struct Params {
    T mass, inv_mass, ixx, iyy, izz, S, b, c;
    T cl_0, cl_alpha, cd_0, cd_alpha2, cl_de, cm_de, cy_dr, cn_dr, cl_da, cm_alpha, cn_beta, cl_beta;
    T damp_roll, damp_pitch, damp_yaw, max_thrust, half_rho, g;
    T min_airspeed, min_u, max_de, max_da, max_dr, thrust_zero_v;
    T max_alpha, max_pitch, max_acc, max_ang_acc;
    T inv_ixx, inv_iyy, inv_izz, sin_max_alpha, cos_max_alpha;
    T half_b, half_c, inv_thrust_zero_v, izz_m_iyy, ixx_m_izz, iyy_m_ixx, half_rho_S, tan_alpha_fast, alpha_needs_atan2,
      sin_max_pitch, cos_max_pitch;   // fp32 evaluation only
    // fp32 evaluation only: the constants the packed roll / yaw moment build-up multiplies by, as the pairs it reads them in
    // (whole members: pairs assembled at the use site from adjacent scalar members made the compiler keep a slice of this
    // struct in scratch)
    fast::f32x2 pk_b_c, pk_half_b_c, pk_damp_pr, pk_beta_ln, pk_inv_i_pr;

    // `blk` points at one FD_NP-word block staged in LDS (stored as double; narrowed here once per launch)
    FD_DEV void load(const double* blk)
    {
        mass = T(blk[FD_P_MASS]); inv_mass = T(blk[FD_PD_INV_MASS]);            // simplified_6dof.py:455
        ixx = T(blk[FD_P_IXX]); iyy = T(blk[FD_P_IYY]); izz = T(blk[FD_P_IZZ]);
        S = T(blk[FD_P_WING_AREA]); b = T(blk[FD_P_WING_SPAN]); c = T(blk[FD_P_CHORD]);
        cl_0 = T(blk[FD_P_CL_0]); cl_alpha = T(blk[FD_P_CL_ALPHA]); cd_0 = T(blk[FD_P_CD_0]);
        cd_alpha2 = T(blk[FD_P_CD_ALPHA2]); cl_de = T(blk[FD_P_CL_ELEVATOR]); cm_de = T(blk[FD_P_CM_ELEVATOR]);
        cy_dr = T(blk[FD_P_CY_RUDDER]); cn_dr = T(blk[FD_P_CN_RUDDER]); cl_da = T(blk[FD_P_CL_AILERON]);
        cm_alpha = T(blk[FD_P_CM_ALPHA]); cn_beta = T(blk[FD_P_CN_BETA]); cl_beta = T(blk[FD_P_CL_BETA]);
        damp_roll = T(blk[FD_P_DAMPING_ROLL]); damp_pitch = T(blk[FD_P_DAMPING_PITCH]);
        damp_yaw = T(blk[FD_P_DAMPING_YAW]); max_thrust = T(blk[FD_P_MAX_THRUST]);
        half_rho = T(0.5 * blk[FD_P_AIR_DENSITY]); g = T(blk[FD_P_GRAVITY]);
        min_airspeed = T(blk[FD_P_MIN_AIRSPEED_AERO]); min_u = T(blk[FD_P_MIN_U_VELOCITY]);
        max_de = T(blk[FD_P_MAX_ELEVATOR_RAD]); max_da = T(blk[FD_P_MAX_AILERON_RAD]);
        max_dr = T(blk[FD_P_MAX_RUDDER_RAD]); thrust_zero_v = T(blk[FD_P_THRUST_ZERO_VELOCITY]);
        max_alpha = T(blk[FD_P_MAX_ALPHA_RAD]); max_pitch = T(blk[FD_P_MAX_PITCH_RAD]);
        max_acc = T(blk[FD_P_MAX_ACCELERATION]); max_ang_acc = T(blk[FD_P_MAX_ANGULAR_ACCELERATION]);
        inv_ixx = T(blk[FD_PD_INV_IXX]); inv_iyy = T(blk[FD_PD_INV_IYY]); inv_izz = T(blk[FD_PD_INV_IZZ]);   // fp32 variants only
        sin_max_alpha = T(blk[FD_PD_SIN_MAX_ALPHA]); cos_max_alpha = T(blk[FD_PD_COS_MAX_ALPHA]);
        half_b = T(0.5) * b; half_c = T(0.5) * c; inv_thrust_zero_v = T(blk[FD_PD_INV_THRUST_ZERO_V]);
        izz_m_iyy = izz - iyy; ixx_m_izz = ixx - izz; iyy_m_ixx = iyy - ixx;
        half_rho_S = half_rho * S; tan_alpha_fast = T(blk[FD_PD_TAN_ALPHA_FAST]);
        alpha_needs_atan2 = T(blk[FD_PD_ALPHA_NEEDS_ATAN2]);
        sin_max_pitch = T(blk[FD_PD_SIN_MAX_PITCH]); cos_max_pitch = T(blk[FD_PD_COS_MAX_PITCH]);
        if constexpr (sizeof(T) == 4) {
            const float b_ = float(blk[FD_P_WING_SPAN]), c_ = float(blk[FD_P_CHORD]);
            pk_b_c = (fast::f32x2){ b_, c_ };
            pk_half_b_c = (fast::f32x2){ 0.5f * b_, 0.5f * c_ };
            pk_damp_pr = (fast::f32x2){ float(blk[FD_P_DAMPING_ROLL]), float(blk[FD_P_DAMPING_YAW]) };
            pk_beta_ln = (fast::f32x2){ float(blk[FD_P_CL_BETA]), float(blk[FD_P_CN_BETA]) };
            pk_inv_i_pr = (fast::f32x2){ float(blk[FD_PD_INV_IXX]), float(blk[FD_PD_INV_IZZ]) };
        }
    }
    // The derived words exist only in the STAGED (LDS) copy of a parameter block (stride FD_NP_STAGED): the first
    // FD_ND_LANES threads of a workgroup fill them while the block is staged (fdyn_kernels.hip: stage_params), one word per
    // lane, instead of every lane of every launch running fp64 ocml sin/cos and divisions before it can start (~400 VALU
    // of a 1600-VALU one-step launch).  Callers of the C-ABI never see them: params stays [n_types][FD_NP].
    //   lanes 0..4 : reciprocals (fp64 division: inv_mass is used by the fp64 parity path too)
    //   lanes 5..6 : sin / cos of the alpha limit and of the pitch limit -- read by the fp32-evaluation variants only, so
    //                computed with the fp32 sincos (25 instructions; the fp64 ocml pair is ~1500 cycles on the path every
    //                wave of the workgroup waits for) and skipped in the fp64 kernels
    static constexpr int FD_ND_LANES = 7;
    // word of the block a derive lane starts from (so that a caller can issue that load EARLY, ahead of its per-aircraft loads)
    static FD_DEV int derive_source(int lane)
    {
        return lane == 0 ? FD_P_MASS : (lane == 1 ? FD_P_IXX : (lane == 2 ? FD_P_IYY : (lane == 3 ? FD_P_IZZ : (lane == 4 ? FD_P_THRUST_ZERO_VELOCITY
               : (lane == 5 ? FD_P_MAX_ALPHA_RAD : FD_P_MAX_PITCH_RAD)))));
    }
    template <bool FAST>
    static FD_DEV void derive_lane(int lane, const double* __restrict__ src, double* blk)
    {
        derive_lane_from<FAST>(lane, src[derive_source(lane < FD_ND_LANES ? lane : 0)], blk);
    }
    template <bool FAST>
    static FD_DEV void derive_lane_from(int lane, double a, double* blk)                 // a = block[derive_source(lane)]
    {
        if (lane < 5) {
            const int to = lane == 0 ? FD_PD_INV_MASS : (lane == 1 ? FD_PD_INV_IXX : (lane == 2 ? FD_PD_INV_IYY : (lane == 3 ? FD_PD_INV_IZZ : FD_PD_INV_THRUST_ZERO_V)));
            blk[to] = 1.0 / a;
        } else if (FAST && lane < FD_ND_LANES) {
            const bool is_alpha = lane == 5;
            float snf, csf;
            fast::sincos(float(a), snf, csf);
            const double sn = snf, cs = csf;
            blk[is_alpha ? FD_PD_SIN_MAX_ALPHA : FD_PD_SIN_MAX_PITCH] = sn;
            blk[is_alpha ? FD_PD_COS_MAX_ALPHA : FD_PD_COS_MAX_PITCH] = cs;
            if (is_alpha) {
                // |w| <= tan(max_alpha) u_safe <=> the alpha clip is inactive; the polynomial serves it while tan <= 0.7
                const double tl = sn / cs;
                const bool poly_ok = a > 0.0 && a < 1.5 && tl <= double(FD_ATAN_WIDE_LIMIT);
                blk[FD_PD_TAN_ALPHA_FAST] = poly_ok ? tl : 0.0;
                blk[FD_PD_ALPHA_NEEDS_ATAN2] = poly_ok ? 0.0 : 1.0;
            }
        }
    }
};

// limits applied to the stored state after each RK4 step (in the storage type)
template <typename S> struct Limits {
    S max_vel, max_rate, max_pitch, min_dt, max_dt;
    FD_DEV void load(const double* blk)
    {
        max_vel = S(blk[FD_P_MAX_VELOCITY]); max_rate = S(blk[FD_P_MAX_RATE_RAD]);
        max_pitch = S(blk[FD_P_MAX_PITCH_RAD]); min_dt = S(blk[FD_P_MIN_TIMESTEP]); max_dt = S(blk[FD_P_MAX_TIMESTEP]);
    }
};

// controls after set_controls' clip (simplified_6dof.py:221-226), pre-multiplied into radians per launch
template <typename T> struct Controls {
    T de_rad, da_rad, dr_rad, throttle;
    // fp32 evaluation only: the control-dependent terms of the coefficient build-up, constant over a launch's sub-steps
    T cl0_de, cm_de, cl_da, cn_dr, cy, thrust_max;
    fast::f32x2 pk_da_dr;                                               // (cl_da, cn_dr) as the pair the moment build-up reads
    template <typename S> FD_DEV void set(const Params<T>& P, S elevator, S aileron, S rudder, S thr)
    {
        de_rad = T(clipv<S>(elevator, S(-1), S(1))) * P.max_de;     // :383
        da_rad = T(clipv<S>(aileron, S(-1), S(1))) * P.max_da;      // :415
        dr_rad = T(clipv<S>(rudder, S(-1), S(1))) * P.max_dr;       // :389
        throttle = T(clipv<S>(thr, S(0), S(1)));
        if constexpr (sizeof(T) == 4) {
            cl0_de = P.cl_0 + P.cl_de * de_rad;                     // :384   CL = cl0 + cl_alpha alpha + cl_de de
            cm_de = P.cm_de * de_rad;                               // :424
            cl_da = P.cl_da * da_rad;                               // :419
            cn_dr = P.cn_dr * dr_rad;                               // :431
            pk_da_dr = (fast::f32x2){ P.cl_da * da_rad, P.cn_dr * dr_rad };
            cy = P.cy_dr * dr_rad;                                  // :390
            thrust_max = P.max_thrust * throttle;                   // :404
        }
    }
};

// ----- one evaluation of the equations of motion: simplified_6dof.py:333-503 ----------------------------
// The fp64 parity form: the reference's operation order, every clamp and guard where the reference has it.
// (The fp32-evaluation variants use dynamics_fast below.)
template <typename T>
FD_DEV void dynamics(const Params<T>& P, const Controls<T>& C, const T (&x)[FD_NX], T (&xd)[FD_NX])
{
    static_assert(sizeof(T) == 8, "fp32 evaluation goes through dynamics_fast");
    const T u = x[3], v = x[4], w = x[5], theta = x[7];
    const T p = x[9], q = x[10], r = x[11];
    T sin_phi, cos_phi, sin_theta, cos_theta, sin_psi, cos_psi;
    M<T>::sincos(x[6], sin_phi, cos_phi);
    M<T>::sincos(theta, sin_theta, cos_theta);
    M<T>::sincos(x[8], sin_psi, cos_psi);

    const T airspeed = M<T>::sqrt(u * u + v * v + w * w);                               // :363
    const T safe_airspeed = pymax(airspeed, P.min_airspeed);                            // :364
    const T au = M<T>::abs(u);
    const T u_safe = au > T(1e-6) ? pymax(au, P.min_u) * signv(u) : P.min_u;            // :368
    T alpha = M<T>::atan2(w, u_safe);
    alpha = clipv(alpha, -P.max_alpha, P.max_alpha);                                    // :370
    T sin_alpha, cos_alpha;
    M<T>::sincos(alpha, sin_alpha, cos_alpha);
    const T beta = M<T>::asin(clipv(v / safe_airspeed, T(-1), T(1)));                   // :376
    const T q_dyn = P.half_rho * (airspeed * airspeed);                                 // :379 (unclamped V)

    const T cl = P.cl_0 + P.cl_alpha * alpha + P.cl_de * C.de_rad;                      // :384-390
    const T cd = P.cd_0 + P.cd_alpha2 * (alpha * alpha);
    const T cy = P.cy_dr * C.dr_rad;
    const T q_S = q_dyn * P.S;
    const T lift = q_S * cl, drag = q_S * cd, side_force = q_S * cy;
    const T fx_aero = -drag * cos_alpha + lift * sin_alpha;                             // :397-399
    const T fz_aero = -drag * sin_alpha - lift * cos_alpha;

    const T thrust_factor = pymax(T(0), T(1) - airspeed / P.thrust_zero_v);             // :403
    const T thrust = P.max_thrust * C.throttle * thrust_factor;

    const T fx = fx_aero + thrust + (-P.g * sin_theta) * P.mass;                        // :409-411
    const T fy = side_force + (P.g * cos_theta * sin_phi) * P.mass;
    const T fz = fz_aero + (P.g * cos_theta * cos_phi) * P.mass;

    const T half_span_over_V = P.b / (T(2) * safe_airspeed);                            // :416-417
    const T half_chord_over_V = P.c / (T(2) * safe_airspeed);
    const T l_moment = q_S * P.b * (P.cl_da * C.da_rad + P.damp_roll * p * half_span_over_V + P.cl_beta * beta);
    const T m_moment = q_S * P.c * (P.cm_de * C.de_rad + P.cm_alpha * alpha + P.damp_pitch * q * half_chord_over_V);
    const T n_moment = q_S * P.b * (P.cn_dr * C.dr_rad + P.damp_yaw * r * half_span_over_V + P.cn_beta * beta);

    const T sps = sin_phi * sin_theta, cps = cos_phi * sin_theta;                       // :440-452
    xd[0] = cos_theta * cos_psi * u + (sps * cos_psi - cos_phi * sin_psi) * v + (cps * cos_psi + sin_phi * sin_psi) * w;
    xd[1] = cos_theta * sin_psi * u + (sps * sin_psi + cos_phi * cos_psi) * v + (cps * sin_psi - sin_phi * cos_psi) * w;
    xd[2] = -sin_theta * u + sin_phi * cos_theta * v + cos_phi * cos_theta * w;

    xd[3] = fx * P.inv_mass - q * w + r * v;                                            // :455-460
    xd[4] = fy * P.inv_mass - r * u + p * w;
    xd[5] = fz * P.inv_mass - p * v + q * u;

    const T theta_safe = clipv(theta, -P.max_pitch, P.max_pitch);                       // :463-471
    const T cos_ts = (theta_safe == theta) ? cos_theta : M<T>::cos(theta_safe);
    const T tan_ts = M<T>::tan(theta_safe);
    xd[8] = (sin_phi * q + cos_phi * r) / cos_ts;
    xd[6] = p + sin_phi * tan_ts * q + cos_phi * tan_ts * r;
    xd[7] = cos_phi * q - sin_phi * r;

    xd[9] = (l_moment - (P.izz - P.iyy) * q * r) / P.ixx;                               // :474-482
    xd[10] = (m_moment - (P.ixx - P.izz) * p * r) / P.iyy;
    xd[11] = (n_moment - (P.iyy - P.ixx) * p * q) / P.izz;

#pragma unroll
    for (int i = 9; i < 12; ++i) xd[i] = clipv(xd[i], -P.max_ang_acc, P.max_ang_acc);  // :485-490
#pragma unroll
    for (int i = 3; i < 6; ++i) xd[i] = clipv(xd[i], -P.max_acc, P.max_acc);
    // :496-501 non-finite derivatives -> 0.  Every derivative is bounded (clamped or O(100)), so their sum is
    // finite iff each one is: one class test on the sum guards a rare, wave-uniformly-skipped fix-up branch.
    T acc = xd[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) acc += xd[i];
    if (!M<T>::finite(acc)) {
#pragma unroll
        for (int i = 0; i < 12; ++i) xd[i] = M<T>::finite(xd[i]) ? xd[i] : T(0);
    }
}

// ----- the same equations for the fp32-evaluation variants ("mixed", "f32"), written for instruction count -----------
// Differences from dynamics<double> (all mathematically neutral; rounding differs at the fp32 ulp level):
//   * sin / cos of the three Euler angles are INPUTS (Trig): the caller carries them and rotates them by the small
//     stage / step increments (fast::rotate_small) instead of evaluating 12 sincos per RK4 step;
//   * north / east rates as Rz(psi) applied to the yaw-free horizontal velocity (9 products instead of 18);
//   * accelerations formed as (aero + thrust) / m + g-terms directly (no multiply by mass and divide again);
//   * control-dependent coefficient terms pre-multiplied per launch (Controls);
//   * the u_safe / alpha-clip / clamped-pitch guards of :364-376,463-471 are a handful of selects on staged constants
//     (sin / cos of the alpha and pitch limits), alpha and beta come from wide no-range-reduction polynomials, and ONE
//     wave-uniformly skipped block holds what is left (side-slip beyond 48 deg, Euler increments too large for the rotation
//     series).  Measured on the bench's random-action fleet: 1.7 % of lane-steps leave the envelope of a
//     narrower first version (|alpha| < 22 deg, |beta| < 30 deg), which put 65 % of the WAVES through its slow block;
//   * no per-evaluation NaN guard: with a finite state every term is finite (divisors are clamped away from 0:
//     V >= min_airspeed, |u_safe| >= min_u, |cos theta_safe| >= cos(max_pitch)), and the state is re-checked after every
//     step by rk4_substeps' combined predicate -- the reference's guard (:496-501) can only fire on a non-finite state,
//     which its own step() (:286-291) never leaves behind either.
struct Trig { fast::f32x2 phi, th, psi; };          // (sin, cos) of roll, pitch, yaw as packed pairs (v_pk_* operands)
FD_DEV Trig trig_of(float phi, float theta, float psi)
{
    Trig t;
    float s, c;
    fast::sincos(phi, s, c); t.phi = (fast::f32x2){ s, c };
    fast::sincos(theta, s, c); t.th = (fast::f32x2){ s, c };
    fast::sincos(psi, s, c); t.psi = (fast::f32x2){ s, c };
    return t;
}
// rotate by (d_phi, d_theta, d_psi); returns max |d|: beyond 0.125 rad the series is not good enough and the next
// dynamics_fast call re-evaluates the sincos in full (its `dmax` argument)
FD_DEV float trig_rotate(const Trig& t0, float dphi, float dth, float dpsi, Trig& t)
{
    t.phi = fast::rotate_small(t0.phi, dphi);
    t.th = fast::rotate_small(t0.th, dth);
    t.psi = fast::rotate_small(t0.psi, dpsi);
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(dphi), __builtin_fabsf(dth)), __builtin_fabsf(dpsi));
}
// the same with the increments h * (k_phi, k_theta), h * k_psi formed here (the first two as one packed product)
FD_DEV float trig_rotate_scaled(const Trig& t0, float h, float kphi, float kth, float kpsi, Trig& t)
{
    const fast::f32x2 d2 = fast::bc(h) * (fast::f32x2){ kphi, kth };
    return trig_rotate(t0, d2.x, d2.y, h * kpsi, t);
}

#define FD_UNLIKELY(c) __builtin_expect(!!(c), 0)
#ifdef FD_PHASE_STAMPS
// timing experiment only: how often wave 0 of a workgroup enters the rare blocks ([0] wave entries, [1] lane entries of the
// dynamics block; [2], [3] the same for the post-step fix-up; [4] lanes rebuilding the trigonometry)
__device__ unsigned fdyn_dbg_cnt[4096 * 8];
#define FD_DBG_COUNT(SLOT) if (threadIdx.x < 64) { const unsigned long long m_ = __ballot(1); \
    if (int(threadIdx.x) == __ffsll((long long)m_) - 1) atomicAdd(&fdyn_dbg_cnt[blockIdx.x * 8 + (SLOT)], 1u); \
    atomicAdd(&fdyn_dbg_cnt[blockIdx.x * 8 + (SLOT) + 1], 1u); }
#define FD_DBG_COUNT1(SLOT) if (threadIdx.x < 64) atomicAdd(&fdyn_dbg_cnt[blockIdx.x * 8 + (SLOT)], 1u);
#else
#define FD_DBG_COUNT(SLOT)
#define FD_DBG_COUNT1(SLOT)
#endif

// x: the 12 state words (x[0..2] unread; x[6], x[8] read only when the trigonometry has to be rebuilt); tg is updated in
// place when it is rebuilt, so a carried Trig stays repaired.
// STRAIGHT: where the STICKY special cases live.  true (launches of at most one wave per SIMD, where the launch lasts as long
// as its slowest wave): in the straight-line path, as selects -- every wave pays ~10 % more instructions, none pays a
// wave-level block on every evaluation.  false (two waves per SIMD, the register-capped env build for batches beyond 65 536:
// throughput-bound, a slow wave's issue gaps are filled by its neighbour): behind the wave-level branches -- measured at 1 Mi
// envs 2.41e9 env-steps/s against 2.25e9 with the straight form.  The arithmetic is the same expression for expression: the
// two forms are bit-equal (tests/test_gpu_parity_scale.py compares the two env builds).
template <bool STRAIGHT = true>
FD_DEV void dynamics_fast(const Params<float>& P, const Controls<float>& C, const float (&x)[FD_NX], Trig& tg, float dmax,
                          float (&xd)[FD_NX])
{
    const float u = x[3], v = x[4], w = x[5], theta = x[7];
    const float p = x[9], q = x[10], r = x[11];

    const fast::f32x2 vw = { v, w };
    const fast::f32x2 vw_sq = vw * vw;                                                      // v^2, w^2
    const float uw2 = __builtin_fmaf(u, u, vw_sq.y);
    const float V2 = __builtin_fmaf(v, v, uw2);
    const float airspeed = fast::sqrt(V2);                                                  // :363
    const float Vs = fast::max_nc(airspeed, P.min_airspeed);                                // :364
    const float inv_V = fast::rcp(Vs);

    // ---- angle of attack :368-370, branch-free.  alpha only matters inside +-max_alpha: |w| <= tan(max_alpha) u_safe (which
    // implies u_safe > 0) <=> the clip is inactive, and then |w / u_safe| <= 0.7 is inside the no-range-reduction
    // polynomial; otherwise atan2(w, u_safe) lies beyond the limit on the side of sign(w) and clips to copysign(max_alpha, w)
    // (u_safe < 0, w = +-0 included: atan2 = +-pi).  sin / cos of the UNCLIPPED alpha are w/h and u_safe/h.
    const float au = __builtin_fabsf(u);
    const float au_min = fast::max_nc(au, P.min_u);
    const float us = au > 1e-6f ? __builtin_copysignf(au_min, u) : P.min_u;
    const float inv_h = fast::rsq(__builtin_fmaf(us, us, vw_sq.y));                         // |u_safe| >= min_u > 0
    const fast::f32x2 xt_ab = vw * (fast::f32x2){ inv_V, fast::rcp(us) };                   // v / V, w / u_safe
    const fast::f32x2 xt_sq = xt_ab * xt_ab;
    const float t_alpha = xt_ab.y;
    const bool a_in = __builtin_fabsf(w) <= P.tan_alpha_fast * us;
    const float sin_alpha = a_in ? w * inv_h : __builtin_copysignf(P.sin_max_alpha, w);
    const float cos_alpha = a_in ? us * inv_h : P.cos_max_alpha;

    // ---- side-slip :376, ONE polynomial, branch-free over the whole range.  Up to |v| / V = 0.75 its argument is v / V; beyond
    // it asin = pi/2 - 2 asin(sqrt((1 - |x|) / 2)) with 1 - |v|/V = (u^2 + w^2) / (V (V + |v|)) -- no cancellation: asin is
    // ill-conditioned at +-1, a tumbling aircraft at the rate clamp flies sideways with |v| / V = 1 - 2e-8, which fp32 rounds to
    // 1 and beta is off by 3e-4 rad (measured as THE source of mixed-precision outliers over 4096 aircraft).  The second form
    // used to live in the rare block below: but sideways flight is STICKY -- in the steady state of the bench's random-action
    // fleet every second wave holds an aircraft beyond 48 deg of side-slip, ran the rare block on all 80 evaluations of an env
    // step (113 k cycles for the RK4 phase against 72 k, scratch/phase_stamps.py) and the launch waited for those waves.
    const float av = __builtin_fabsf(v);
    const bool b_in = av <= FD_ASIN_WIDE_LIMIT * Vs;
    const float xb = xt_ab.x;
    auto beta_tail = [&]() {                                 // |v| / V beyond 0.75
        const float vv_rest = airspeed >= P.min_airspeed ? uw2 : __builtin_fmaxf(__builtin_fmaf(Vs, Vs, -(v * v)), 0.0f);   // V^2 - v^2 (V clamped: :364)
        const float half_om = 0.5f * (vv_rest * fast::rcp(Vs * (Vs + av)));                 // (1 - |v| / V) / 2 <= 0.125
        const float b_r = fast::asin_wide_t(fast::sqrt(half_om), half_om);
        return __builtin_copysignf(__builtin_fmaf(-2.0f, b_r, 1.5707963267948966f), v);
    };
    // atan_wide(t_alpha) and asin_wide_t(b_t, z_b) -- the two no-range-reduction polynomials -- share their last four Horner
    // steps, the p z product and the final FMA as packed instructions (the same FMAs, two per instruction)
    float beta, b_t, z_b;
    if constexpr (STRAIGHT) {
        const float vv_rest = airspeed >= P.min_airspeed ? uw2 : __builtin_fmaxf(__builtin_fmaf(Vs, Vs, -vw_sq.x), 0.0f);
        const float half_om = 0.5f * (vv_rest * fast::rcp(Vs * (Vs + av)));
        b_t = b_in ? xb : fast::sqrt(half_om);
        z_b = b_in ? xt_sq.x : half_om;
    } else {
        b_t = xb; z_b = xt_sq.x;                            // the tail is fixed up in the rare block below
    }
    const fast::f32x2 ab = fast::atan_asin_wide(t_alpha, xt_sq.y, b_t, z_b);
    const float alpha_poly = fast::pinned(ab.x);
    const float alpha = a_in ? alpha_poly : __builtin_copysignf(P.max_alpha, w);
    if constexpr (STRAIGHT) beta = b_in ? ab.y : __builtin_copysignf(__builtin_fmaf(-2.0f, ab.y, 1.5707963267948966f), v);
    else beta = ab.y;

    // ---- clamped pitch for the Euler rates :463: sin / cos of clip(theta) are the carried ones or those of +-max_pitch
    const bool th_in = __builtin_fabsf(theta) <= P.max_pitch;
    float sth_e = th_in ? tg.th.x : __builtin_copysignf(P.sin_max_pitch, theta);
    float cth_e = th_in ? tg.th.y : P.cos_max_pitch;

    // ---- the rare block: an Euler-angle increment too large for the rotation series (after a wrap / pitch clamp, or 10 ms steps
    // of a tumbling aircraft), or an aircraft type whose alpha limit lies beyond the polynomial -- transient or absent
    const bool ordinary = (dmax <= 0.125f) & (P.alpha_needs_atan2 == 0.0f) & (STRAIGHT | b_in);
    float alpha_r = alpha, sin_alpha_r = sin_alpha, cos_alpha_r = cos_alpha;
    if (FD_UNLIKELY(!ordinary)) {
        FD_DBG_COUNT(0)
        if (!(dmax <= 0.125f)) {
            FD_DBG_COUNT1(4)
            tg = trig_of(x[6], theta, x[8]);
            sth_e = th_in ? tg.th.x : sth_e; cth_e = th_in ? tg.th.y : cth_e;
        }
        if (P.alpha_needs_atan2 != 0.0f) {                  // max_alpha > 35 deg: the reference's own sequence
            const float a_raw = fast::atan2(w, us);
            const bool hi = a_raw > P.max_alpha, lo = a_raw < -P.max_alpha;
            sin_alpha_r = hi ? P.sin_max_alpha : (lo ? -P.sin_max_alpha : w * inv_h);
            cos_alpha_r = (hi || lo) ? P.cos_max_alpha : us * inv_h;
            alpha_r = clipf(a_raw, -P.max_alpha, P.max_alpha);
        }
        if constexpr (!STRAIGHT) {
            if (!b_in) beta = beta_tail();
        }
    }
    const float sth = tg.th.x, cth = tg.th.y;

    using fast::f32x2;
    using fast::bc;
    const float q_S = P.half_rho_S * V2;                                                    // :379 (unclamped V)
    const float cl = __builtin_fmaf(P.cl_alpha, alpha_r, C.cl0_de);                           // :384-390
    const float cd = __builtin_fmaf(P.cd_alpha2, alpha_r * alpha_r, P.cd_0);
    const f32x2 ld = bc(q_S) * (f32x2){ cl, cd };                                           // lift, drag
    // :397-399  (fx, -fz) = (lift sin - drag cos, lift cos + drag sin): a plane rotation, one packed product + one packed FMA
    const f32x2 cs_a = { cos_alpha_r, sin_alpha_r };
    const f32x2 fxz = fast::pk_fma_ayx_bx_nlo(cs_a, ld, ld.yy * cs_a);
    const float thrust = C.thrust_max * __builtin_fmaxf(0.0f, __builtin_fmaf(-airspeed, P.inv_thrust_zero_v, 1.0f));   // :403

    // :409-411 + :455-460   a = F/m + g-terms - omega x v
    const f32x2 g_cs = bc(P.g) * (tg.phi * bc(cth));                                        // g (cth sphi), g (cth cphi)
    xd[3] = __builtin_fmaf(fxz.x + thrust, P.inv_mass, __builtin_fmaf(r, v, __builtin_fmaf(-q, w, -(P.g * sth))));
    xd[4] = __builtin_fmaf(q_S * C.cy, P.inv_mass, __builtin_fmaf(p, w, __builtin_fmaf(-r, u, g_cs.x)));
    xd[5] = __builtin_fmaf(-fxz.y, P.inv_mass, __builtin_fmaf(q, u, __builtin_fmaf(-p, v, g_cs.y)));

    // :416-431, :474-482  roll and yaw moments have one shape: packed as (l, n)
    const f32x2 hV = P.pk_half_b_c * bc(inv_V);                             // :416-417
    const f32x2 qSbc = bc(q_S) * P.pk_b_c;
    const f32x2 pr = { p, r };
    const f32x2 ln_in = fast::pk_fma(P.pk_damp_pr * pr, bc(hV.x), C.pk_da_dr);
    const f32x2 ln = bc(qSbc.x) * fast::pk_fma(P.pk_beta_ln, bc(beta), ln_in);
    const float m_moment = qSbc.y * __builtin_fmaf(P.cm_alpha, alpha_r, __builtin_fmaf(P.damp_pitch * q, hV.y, C.cm_de));
    const f32x2 gy = { __builtin_fmaf(-P.izz_m_iyy * q, r, ln.x), __builtin_fmaf(-P.iyy_m_ixx * p, q, ln.y) };
    const f32x2 a_pr = gy * P.pk_inv_i_pr;                                // :474-482
    xd[9] = a_pr.x;
    xd[10] = __builtin_fmaf(-P.ixx_m_izz * p, r, m_moment) * P.inv_iyy;
    xd[11] = a_pr.y;

    // :440-452 NED rates: Rz(psi) * [ (cth u + sth (sphi v + cphi w)), (cphi v - sphi w) ],  down = cth (sphi v + cphi w) - sth u
    // the plane rotations as packed pairs (the scalar form's products and FMAs, two per instruction):
    // (sv_cw, bh) = (sphi v + cphi w, cphi v - sphi w)
    const fast::f32x2 sb = fast::pk_fma_bx_nhi(tg.phi, fast::lo_only(v), tg.phi.yx * fast::bc(w));
    const float sv_cw = sb.x;
    const float ah = __builtin_fmaf(cth, u, sth * sv_cw);
    // (north, east) = (cpsi ah - spsi bh, spsi ah + cpsi bh)
    const fast::f32x2 ne = fast::pk_fma_ayx_bx_nlo(tg.psi, fast::lo_only(ah), tg.psi * fast::bc(sb.y));
    xd[0] = ne.x;
    xd[1] = ne.y;
    xd[2] = __builtin_fmaf(cth, sv_cw, -(sth * u));

    // :463-471 Euler rates with the clamped pitch:  (qr, theta_dot) = (sphi q + cphi r, cphi q - sphi r)
    const float inv_c = fast::rcp(cth_e);
    const fast::f32x2 qt = fast::pk_fma_bx_nhi(tg.phi, fast::lo_only(q), tg.phi.yx * fast::bc(r));
    const float qr = qt.x;
    xd[8] = qr * inv_c;
    xd[6] = __builtin_fmaf(sth_e * inv_c, qr, p);
    xd[7] = qt.y;

#pragma unroll
    for (int i = 9; i < 12; ++i) xd[i] = clipf(xd[i], -P.max_ang_acc, P.max_ang_acc);      // :485-490
#pragma unroll
    for (int i = 3; i < 6; ++i) xd[i] = clipf(xd[i], -P.max_acc, P.max_acc);
}

// roll / yaw re-wrap of the stored state (simplified_6dof.py:266,270).  The f64 variant keeps the
// reference's atan2(sin, cos) form; the fp32-compute variants use the equivalent range reduction.
template <typename S, typename T> FD_DEV S wrap_state_angle(S a)
{
    if constexpr (sizeof(T) == 8) {
        S s, c;
        M<S>::sincos(a, s, c);
        return M<S>::atan2(s, c);
    } else {
        return a - S(2.0 * FD_PI) * M<S>::rint(a * S(1.0 / (2.0 * FD_PI)));
    }
}

// ----- post-step clamps of Simplified6DOF.step, simplified_6dof.py:256-291 (in the storage type) -------------
template <typename S, typename T>
FD_DEV void post_step(const Limits<S>& Lm, S (&x)[FD_NX])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                                      // :258 nan_to_num
        const S v = x[i];
        x[i] = (v != v) ? S(0) : (M<S>::finite(v) ? v : (v > S(0) ? S(10000) : S(-10000)));
    }
#pragma unroll
    for (int i = 3; i < 6; ++i) x[i] = clipv(x[i], -Lm.max_vel, Lm.max_vel);          // :262
    x[6] = wrap_state_angle<S, T>(x[6]);                                               // :266
    x[7] = clipv(x[7], -Lm.max_pitch, Lm.max_pitch);                                   // :268
    x[8] = wrap_state_angle<S, T>(x[8]);                                               // :270
#pragma unroll
    for (int i = 9; i < 12; ++i) x[i] = clipv(x[i], -Lm.max_rate, Lm.max_rate);       // :273
    if (-x[2] < S(0)) {                                                                // :276-283 ground clamp
        x[2] = S(0);
        x[5] = x[5] > S(0) ? x[5] : S(0);
    }
    S accs = x[0];                                                                     // :286-291: all finite <=> sum finite
#pragma unroll
    for (int i = 1; i < 12; ++i) accs += x[i];
    if (!M<S>::finite(accs)) {
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] = M<S>::finite(x[i]) ? x[i] : S(0);
    }
}

// ----- the fp32-evaluation integrator state: fp32 copy of the stored state + carried trigonometry -------------------
// Lives in registers across the sub-steps of a launch AND across the control steps of the cascade kernels, whose glue
// (derived scalars, guidance) reads the same fp32 copy and the same sin / cos.
struct FastRK {
    float x0[FD_NX];        // fp32 copy of the stored state
    Trig t0;                // sin / cos of x0[6..8], rotated incrementally
    float d0;               // max |increment| t0 was last rotated by (> 0.125: the next user rebuilds it in full)
    template <typename S> FD_DEV void init(const S (&x)[FD_NX])
    {
#pragma unroll
        for (int i = 0; i < 12; ++i) x0[i] = float(x[i]);
        t0 = trig_of(x0[6], x0[7], x0[8]);
        d0 = 0.0f;
    }
    FD_DEV void ensure_trig()
    {
        if (FD_UNLIKELY(!(d0 <= 0.125f))) { t0 = trig_of(x0[6], x0[7], x0[8]); d0 = 0.0f; }
    }
};

// one RK4 step of `dt` in fp32 evaluation: the four stages and their weighted sum run entirely in fp32 from the fp32 copy of
// the state; the storage type S sees ONE add per word per step (x += S(dt/6 * sum)) -- that add is what keeps the "mixed"
// variant inside the 1e-4 gate.  The clamps / wraps of :256-291 are tested on the fp32 copy with one combined predicate and
// the (rare) fix-up runs under a wave-level branch.
template <typename S, bool STRAIGHT = true>
FD_DEV void rk4_fast_step(const Params<float>& P, const Limits<S>& Lm, const Controls<float>& C, S (&x)[FD_NX], FastRK& f,
                          float hdt, float fdt, float dt6)
{
    using T = float;
    // position (0..2) feeds nothing back, and roll / yaw enter only through their sin / cos: the stage states carry velocity,
    // angles (for the rare full rebuild and the +-85 deg guard) and rates; the trigonometry is rotated
    T xt[FD_NX], k[FD_NX];
    Trig tt;
    // the stage combinations as packed pairs (words PA[j], PB[j]): acc += w k and xt = x0 + h k are two FMAs per instruction.
    // (Pairing (8,10) (9,11), so that (p, r) -- the pair the roll / yaw moment build-up reads and writes -- passes through
    // whole, measured four instructions MORE per sub-step than the plain order: the storage-type accumulate hands the words
    // back one by one either way.)
    using fast::f32x2;
    constexpr int PA[6] = { 0, 2, 4, 6, 8, 10 }, PB[6] = { 1, 3, 5, 7, 9, 11 };
    f32x2 acc2[6], x02[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) x02[j] = (f32x2){ f.x0[PA[j]], f.x0[PB[j]] };
    auto stage = [&](T h, T wgt, bool first) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x2 kj = { k[PA[j]], k[PB[j]] };
            acc2[j] = first ? kj : fast::pk_fma(fast::bc(wgt), kj, acc2[j]);
            if (j >= 1) {                                           // words 0..2 feed nothing back (word 2 rides along in its pair)
                const f32x2 xj = fast::pk_fma(fast::bc(h), kj, x02[j]);
                xt[PA[j]] = xj.x; xt[PB[j]] = xj.y;
            }
        }
    };
    dynamics_fast<STRAIGHT>(P, C, f.x0, f.t0, f.d0, k);                            // k1
    stage(hdt, T(1), true);
    T dm = trig_rotate_scaled(f.t0, hdt, k[6], k[7], k[8], tt);
    dynamics_fast<STRAIGHT>(P, C, xt, tt, dm, k);                        // k2
    stage(hdt, T(2), false);
    dm = trig_rotate_scaled(f.t0, hdt, k[6], k[7], k[8], tt);
    dynamics_fast<STRAIGHT>(P, C, xt, tt, dm, k);                        // k3
    stage(fdt, T(2), false);
    dm = trig_rotate_scaled(f.t0, fdt, k[6], k[7], k[8], tt);
    dynamics_fast<STRAIGHT>(P, C, xt, tt, dm, k);                        // k4
    T ksum[FD_NX], inc[FD_NX];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x2 ks = acc2[j] + (f32x2){ k[PA[j]], k[PB[j]] };
        const f32x2 in2 = fast::bc(dt6) * ks;
        ksum[PA[j]] = ks.x; ksum[PB[j]] = ks.y;
        inc[PA[j]] = in2.x; inc[PB[j]] = in2.y;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        // fp32 storage: the accumulate is spelled as the one fused instruction, so that no build's contraction heuristics decide
        if constexpr (sizeof(S) == 4) x[i] = __builtin_fmaf(dt6, ksum[i], x[i]);
        else x[i] += S(inc[i]);
        // the body-rate clamp (:273) sits in the straight-line path, as a select in the storage type: it is the STICKY clamp -- an
        // aircraft tumbling against it trips it on every sub-step, and behind the wave-level branch below it cost the wave
        // holding that aircraft 20 fix-ups per env step (at one wave per SIMD the launch lasts as long as its slowest wave)
        if (STRAIGHT && i >= 9) x[i] = M<S>::abs(x[i]) > Lm.max_rate ? M<S>::copysign(Lm.max_rate, x[i]) : x[i];
        f.x0[i] = T(x[i]);
    }
    f.d0 = trig_rotate(f.t0, inc[6], inc[7], inc[8], f.t0);
    // one predicate for every clamp / wrap / guard of :256-291, evaluated on the fp32 copy
    f32x2 sum2 = { f.x0[0], f.x0[1] };                   // all finite <=> the sum is finite: only its finiteness is read
#pragma unroll
    for (int j = 1; j < 6; ++j) sum2 += (f32x2){ f.x0[2 * j], f.x0[2 * j + 1] };
    const T sum = sum2.x + sum2.y;
    const T vmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(f.x0[3]), __builtin_fabsf(f.x0[4])), __builtin_fabsf(f.x0[5]));
    const T amax = __builtin_fmaxf(__builtin_fabsf(f.x0[6]), __builtin_fabsf(f.x0[8]));
    // Two classes.  `lim`: the velocity clamp, ground contact and the roll / yaw re-wraps (:262, :266, :270, :276-283) leave
    // sin / cos of the angles alone (a wrap moves the angle by 2 pi), so the carried trigonometry stays valid and only the
    // touched words are refreshed -- in the steady state of the bench's fleet some aircraft of every wave wraps its yaw once
    // per env step.  `ang`: the pitch clamp and non-finite values; these take the full post_step and ask for the rebuild.  (Round 2's first form put all of post_step + a full
    // sincos rebuild behind one predicate that included the rate clamp: a wave holding one aircraft at that clamp ran 20
    // fix-ups + 19 rebuilds per env step, 100-111 k cycles for the RK4 phase against 71.6 k for an undisturbed wave --
    // scratch/phase_stamps.py, shader-clock stamps -- and the launch waited for it.)
    bool lim = !(vmax <= T(Lm.max_vel)) | (f.x0[2] > T(0)) | !(amax <= T(FD_PI));
    if constexpr (!STRAIGHT) {
        // the same comparison the straight form makes, in the storage type (an fp32 copy can sit exactly on the limit while
        // the stored value is a rounding beyond it)
        lim |= (M<S>::abs(x[9]) > Lm.max_rate) | (M<S>::abs(x[10]) > Lm.max_rate) | (M<S>::abs(x[11]) > Lm.max_rate);
    }
    const bool ang = !(__builtin_fabsf(f.x0[7]) <= T(Lm.max_pitch)) | !M<T>::finite(sum);
    if (FD_UNLIKELY(lim | ang)) {
        FD_DBG_COUNT(2)
#ifdef FD_PHASE_STAMPS
        if (!(vmax <= T(Lm.max_vel))) { FD_DBG_COUNT1(5) }
        if (f.x0[2] > T(0)) { FD_DBG_COUNT1(7) }
#endif
        if (ang) {
            post_step<S, T>(Lm, x);
#pragma unroll
            for (int i = 0; i < 12; ++i) f.x0[i] = T(x[i]);
            f.d0 = T(1);                                         // the next user rebuilds the trigonometry in full
        } else {
#pragma unroll
            for (int i = 3; i < 6; ++i) x[i] = clipv(x[i], -Lm.max_vel, Lm.max_vel);          // :262
            if constexpr (!STRAIGHT) {
#pragma unroll
                for (int i = 9; i < 12; ++i) {                                                 // :273
                    x[i] = M<S>::abs(x[i]) > Lm.max_rate ? M<S>::copysign(Lm.max_rate, x[i]) : x[i];
                    f.x0[i] = T(x[i]);
                }
            }
            x[6] = wrap_state_angle<S, T>(x[6]);                                               // :266
            x[8] = wrap_state_angle<S, T>(x[8]);                                               // :270
            if (-x[2] < S(0)) {                                                                // :276-283 ground clamp
                x[2] = S(0);
                x[5] = x[5] > S(0) ? x[5] : S(0);
            }
            f.x0[2] = T(x[2]); f.x0[6] = T(x[6]); f.x0[8] = T(x[8]);
#pragma unroll
            for (int i = 3; i < 6; ++i) f.x0[i] = T(x[i]);
        }
    }
}

// ----- Simplified6DOF.step x n_sub: RK4 + post-clamps, simplified_6dof.py:247-291 ------------------------
// fp64 evaluation (T = double): the reference's operation order, every clamp applied every step.
// fp32 evaluation (T = float): rk4_fast_step above.
template <typename S, typename T, bool STRAIGHT = true>
FD_DEV void rk4_substeps(const Params<T>& P, const Limits<S>& Lm, const Controls<T>& C, S (&x)[FD_NX], S dt, int n_sub)
{
    if constexpr (sizeof(T) == 8) {
        for (int s = 0; s < n_sub; ++s) {
            T xt[FD_NX], k[FD_NX];
            S acc[FD_NX];
            const T hdt = T(S(0.5) * dt), fdt = T(dt);
#pragma unroll
            for (int i = 0; i < 12; ++i) xt[i] = T(x[i]);
            dynamics<T>(P, C, xt, k);                                        // k1
#pragma unroll
            for (int i = 0; i < 12; ++i) { acc[i] = S(k[i]); xt[i] = T(x[i] + S(hdt) * S(k[i])); }
            dynamics<T>(P, C, xt, k);                                        // k2
#pragma unroll
            for (int i = 0; i < 12; ++i) { acc[i] += S(2) * S(k[i]); xt[i] = T(x[i] + S(hdt) * S(k[i])); }
            dynamics<T>(P, C, xt, k);                                        // k3
#pragma unroll
            for (int i = 0; i < 12; ++i) { acc[i] += S(2) * S(k[i]); xt[i] = T(x[i] + S(fdt) * S(k[i])); }
            dynamics<T>(P, C, xt, k);                                        // k4
            const S dt6 = dt / S(6);
#pragma unroll
            for (int i = 0; i < 12; ++i) x[i] = x[i] + dt6 * (acc[i] + S(k[i]));              // :253
            post_step<S, T>(Lm, x);
        }
    } else {
        FastRK f;
        f.init(x);                                               // the ONLY full sincos of the launch (rare blocks aside)
        const T hdt = T(S(0.5) * dt), fdt = T(dt), dt6 = T(dt / S(6));
        for (int s = 0; s < n_sub; ++s) rk4_fast_step<S, STRAIGHT>(P, Lm, C, x, f, hdt, fdt, dt6);
    }
}

// ----- get_state's derived scalars: simplified_6dof.py:295-331 + _body_to_ned :505-530 -------------------
template <typename S> struct Derived { S airspeed, altitude, ground_speed, heading; };
template <typename S> FD_DEV Derived<S> derived(const S (&x)[FD_NX])
{
    Derived<S> d;
    const S u = x[3], v = x[4], w = x[5];
    S sphi, cphi, sth, cth, spsi, cpsi;
    M<S>::sincos(x[6], sphi, cphi);
    M<S>::sincos(x[7], sth, cth);
    M<S>::sincos(x[8], spsi, cpsi);
    d.airspeed = M<S>::sqrt(u * u + v * v + w * w);
    d.altitude = -x[2];
    const S vn = (cth * cpsi) * u + (sphi * sth * cpsi - cphi * spsi) * v + (cphi * sth * cpsi + sphi * spsi) * w;
    const S ve = (cth * spsi) * u + (sphi * sth * spsi + cphi * cpsi) * v + (cphi * sth * spsi - sphi * cpsi) * w;
    d.heading = M<S>::atan2(ve, vn);
    d.ground_speed = M<S>::sqrt(vn * vn + ve * ve);
    return d;
}
// the same four scalars from the integrator's fp32 copy and its carried sin / cos (no sincos at all): the fp32-evaluation
// cascade's glue.  v_NED = Rz(psi) [cth u + sth (sphi v + cphi w), cphi v - sphi w] as in dynamics_fast.
FD_DEV Derived<float> derived_fast(FastRK& f)
{
    f.ensure_trig();
    Derived<float> d;
    const float u = f.x0[3], v = f.x0[4], w = f.x0[5];
    const Trig& t = f.t0;
    d.airspeed = fast::sqrt(__builtin_fmaf(u, u, __builtin_fmaf(v, v, w * w)));
    d.altitude = -f.x0[2];
    const float sv_cw = __builtin_fmaf(t.phi.x, v, t.phi.y * w);
    const float ah = __builtin_fmaf(t.th.y, u, t.th.x * sv_cw);
    const float bh = __builtin_fmaf(t.phi.y, v, -(t.phi.x * w));
    const float vn = __builtin_fmaf(t.psi.y, ah, -(t.psi.x * bh)), ve = __builtin_fmaf(t.psi.x, ah, t.psi.y * bh);
    d.heading = fast::atan2(ve, vn);
    d.ground_speed = fast::sqrt(__builtin_fmaf(vn, vn, ve * ve));
    return d;
}
// airspeed / altitude only (what the env needs every step)
template <typename S> FD_DEV void airspeed_altitude(const S (&x)[FD_NX], S& airspeed, S& altitude)
{
    airspeed = M<S>::sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
    altitude = -x[2];
}

// ----- fp32 PID: cpp/src/pid_controller.cpp:24-60, bit-faithful (no FMA contraction, same op order) ------
struct PidCfg { float kp, ki, kd, out_min, out_max, int_min, int_max, alpha; };
struct PidState { float integral, err_prev, dfilt; };

FD_DEV float pid_clamp(float v, float lo, float hi)
{   // std::max(min_val, std::min(value, max_val))   pid_controller.cpp:79-81
    const float m = (hi < v) ? hi : v;
    return (lo < m) ? m : lo;
}

FD_DEV float pid_compute(const PidCfg& c, PidState& s, float setpoint, float measurement, float dt)
{
#pragma clang fp contract(off)
    const float error = setpoint - measurement;
    const float p_term = c.kp * error;
    float integral = s.integral + error * dt;
    integral = pid_clamp(integral, c.int_min, c.int_max);
    const float i_term = c.ki * integral;
    const float derivative = (dt > 1e-6f) ? (error - s.err_prev) / dt : 0.0f;
    const float dfilt = c.alpha * derivative + (1.0f - c.alpha) * s.dfilt;
    const float d_term = c.kd * dfilt;
    float out = p_term + i_term + d_term;
    out = pid_clamp(out, c.out_min, c.out_max);
    s.integral = integral; s.err_prev = error; s.dfilt = dfilt;
    return out;
}

// The same controller for the fp32-evaluation cascade (glue type float): FMA contraction allowed, v_med3 clamps, one
// reciprocal of dt per launch instead of nine IEEE divisions per control step (~35 -> ~13 VALU per PID, none of the
// v_cmp -> v_cndmask hazard pairs).  Its inputs already carry fp32 rounding of the state, so it differs from the bit-faithful
// form by the same order (an ulp of the output); the fp64 parity variant, the fused rate-PID demonstrator of the env kernel
// and fdyn_pid_compute_batch keep pid_compute.
FD_DEV float pid_compute_fast(const PidCfg& c, PidState& s, float setpoint, float measurement, float dt, float inv_dt)
{
    const float error = setpoint - measurement;
    const float integral = clipf(__builtin_fmaf(error, dt, s.integral), c.int_min, c.int_max);
    const float derivative = (error - s.err_prev) * inv_dt;                  // inv_dt = 0 when dt <= 1e-6 (pid_controller.cpp:41)
    const float dfilt = __builtin_fmaf(c.alpha, derivative, (1.0f - c.alpha) * s.dfilt);
    const float out = clipf(__builtin_fmaf(c.kp, error, __builtin_fmaf(c.ki, integral, c.kd * dfilt)), c.out_min, c.out_max);
    s.integral = integral; s.err_prev = error; s.dfilt = dfilt;
    return out;
}
// dispatch on the glue type: double -> bit-faithful, float -> contracted
template <typename G> struct PidDt {
    float dt, inv_dt;
    FD_DEV explicit PidDt(G d) : dt(float(d)), inv_dt(float(d) > 1e-6f ? fast::rcp(float(d)) : 0.0f) {}
    FD_DEV float run(const PidCfg& c, PidState& s, float sp, float meas) const
    {
        if constexpr (sizeof(G) == 8) return pid_compute(c, s, sp, meas, dt);
        else return pid_compute_fast(c, s, sp, meas, dt, inv_dt);
    }
};

FD_DEV PidCfg load_pid_cfg(const float* t, int which)
{
    const float* r = t + which * FD_NPC;
    return PidCfg{ r[FD_PC_KP], r[FD_PC_KI], r[FD_PC_KD], r[FD_PC_OUT_MIN], r[FD_PC_OUT_MAX],
                   r[FD_PC_INT_MIN], r[FD_PC_INT_MAX], r[FD_PC_ALPHA] };
}

// ----- the cascade (glue in G = storage type, PIDs fp32) ----------------------------------------------------
template <typename G> struct Surfaces { G elevator, aileron, rudder, throttle; };

struct CascadePids { PidCfg cfg[FD_NPID]; };
struct CascadeState { PidState s[FD_NPID]; };

// controllers/rate_agent.py:92-122
template <typename G>
FD_DEV Surfaces<G> rate_agent(const PidCfg* cfg, PidState* st, const G* C, G p_cmd, G q_cmd, G r_cmd, G throttle,
                              const G (&x)[FD_NX], G dt)
{
    p_cmd = clipv(p_cmd, -C[FD_C_MAX_ROLL_RATE], C[FD_C_MAX_ROLL_RATE]);
    q_cmd = clipv(q_cmd, -C[FD_C_MAX_PITCH_RATE], C[FD_C_MAX_PITCH_RATE]);
    r_cmd = clipv(r_cmd, -C[FD_C_MAX_YAW_RATE], C[FD_C_MAX_YAW_RATE]);
    const PidDt<G> pd(dt);
    const float o_roll = pd.run(cfg[FD_PID_RATE_ROLL], st[FD_PID_RATE_ROLL], float(p_cmd), float(x[9]));
    const float o_pitch = pd.run(cfg[FD_PID_RATE_PITCH], st[FD_PID_RATE_PITCH], float(q_cmd), float(x[10]));
    const float o_yaw = pd.run(cfg[FD_PID_RATE_YAW], st[FD_PID_RATE_YAW], float(r_cmd), float(x[11]));
    Surfaces<G> s;
    s.aileron = clipv(G(o_roll), G(-1), G(1));
    s.elevator = clipv(-G(o_pitch), G(-1), G(1));      // INVERTED (rate_agent.py:111-112)
    s.rudder = clipv(-G(o_yaw), G(-1), G(1));
    s.throttle = clipv(throttle, G(0), G(1));
    return s;
}

// controllers/attitude_agent.py:116-152
template <typename G>
FD_DEV Surfaces<G> attitude_agent(const PidCfg* cfg, PidState* st, const G* C, G roll_cmd, G pitch_cmd, G yaw_cmd,
                                  bool has_yaw, G throttle, const G (&x)[FD_NX], G dt)
{
    roll_cmd = clipv(roll_cmd, -C[FD_C_MAX_ROLL], C[FD_C_MAX_ROLL]);
    pitch_cmd = clipv(pitch_cmd, -C[FD_C_MAX_PITCH], C[FD_C_MAX_PITCH]);
    yaw_cmd = has_yaw ? wrap_angle<G>(yaw_cmd) : G(0);
    const G cur_yaw = wrap_angle<G>(x[8]);
    const PidDt<G> pd(dt);
    const float o_r = pd.run(cfg[FD_PID_ATT_ROLL], st[FD_PID_ATT_ROLL], float(roll_cmd), float(x[6]));
    const float o_p = pd.run(cfg[FD_PID_ATT_PITCH], st[FD_PID_ATT_PITCH], float(pitch_cmd), float(x[7]));
    const float o_y = pd.run(cfg[FD_PID_ATT_YAW], st[FD_PID_ATT_YAW], float(yaw_cmd), float(cur_yaw));
    const G p_cmd = clipv(G(o_r), -C[FD_C_MAX_ROLL_RATE], C[FD_C_MAX_ROLL_RATE]);
    const G q_cmd = clipv(G(o_p), -C[FD_C_MAX_PITCH_RATE], C[FD_C_MAX_PITCH_RATE]);
    const G r_cmd = clipv(G(o_y), -C[FD_C_MAX_YAW_RATE], C[FD_C_MAX_YAW_RATE]);
    return rate_agent<G>(cfg, st, C, p_cmd, q_cmd, r_cmd, throttle, x, dt);
}

// controllers/hsa_agent.py:149-229
template <typename G>
FD_DEV Surfaces<G> hsa_agent(const PidCfg* cfg, PidState* st, const G* C, G heading_cmd, G speed_cmd, G altitude_cmd,
                             const G (&x)[FD_NX], const Derived<G>& d, G dt)
{
    const G heading_error = wrap_angle<G>(heading_cmd - d.heading);
    const G virtual_setpoint = d.heading + heading_error;
    const PidDt<G> pd(dt);
    G roll_angle = G(pd.run(cfg[FD_PID_HEADING], st[FD_PID_HEADING], float(virtual_setpoint), float(d.heading)));
    roll_angle = clipv(roll_angle, -C[FD_C_MAX_BANK_RAD], C[FD_C_MAX_BANK_RAD]);

    const G g = G(9.81), h = d.altitude, V = d.airspeed;                     // :173-185
    const G E_specific = g * h + G(0.5) * (V * V);
    const G E_specific_cmd = g * altitude_cmd + G(0.5) * (speed_cmd * speed_cmd);
    const G E_balance = g * h - G(0.5) * (V * V);
    const G E_balance_cmd = g * altitude_cmd - G(0.5) * (speed_cmd * speed_cmd);

    const G thr_adj = G(pd.run(cfg[FD_PID_ENERGY], st[FD_PID_ENERGY], float(E_specific_cmd), float(E_specific)));
    G throttle = clipv(C[FD_C_BASELINE_THROTTLE] + thr_adj, G(0), G(1));

    G pitch_angle = G(pd.run(cfg[FD_PID_BALANCE], st[FD_PID_BALANCE], float(E_balance_cmd), float(E_balance)));
    G cos_roll;                                                              // :205-208
    if constexpr (sizeof(G) == 8) cos_roll = M<G>::cos(roll_angle);
    else cos_roll = fast::cos_bounded(roll_angle);
    if (M<G>::abs(cos_roll) > G(0.01)) {
        const G load_factor = fdiv(G(1), cos_roll);
        pitch_angle += C[FD_C_LOAD_FACTOR_GAIN] * (load_factor - G(1));
    }
    pitch_angle = clipv(pitch_angle, -C[FD_C_MAX_PITCH_CMD_RAD], C[FD_C_MAX_PITCH_CMD_RAD]);
    return attitude_agent<G>(cfg, st, C, roll_angle, pitch_angle, x[8], true, throttle, x, dt);
}

template <typename G> FD_DEV G wrap_pi(G a)
{   // np.arctan2(np.sin(a), np.cos(a))
    if constexpr (sizeof(G) == 4) {
        return __builtin_fmaf(-6.2831853071795865f, __builtin_rintf(a * 0.15915494309189535f), a);
    } else {
        G s, c;
        M<G>::sincos(a, s, c);
        return M<G>::atan2(s, c);
    }
}

// controllers/waypoint_agent.py:107-242 ; wp = {north, east, altitude, speed}
template <typename G>
FD_DEV Surfaces<G> waypoint_agent(const PidCfg* cfg, PidState* st, const G* C, const G* wp, const G (&x)[FD_NX],
                                  const Derived<G>& d, G dt)
{
    const G e0 = wp[FD_WP_NORTH] - x[0], e1 = wp[FD_WP_EAST] - x[1];
    const G hd = M<G>::sqrt(e0 * e0 + e1 * e1);
    const int gtype = int(C[FD_C_GUIDANCE_TYPE]);
    const G los = M<G>::atan2(e1, e0);
    G heading_cmd = los;
    if (gtype == FD_GUIDANCE_LOS) {                                          // :118-144
        const G V = pymax(d.airspeed, G(10));
        G turn_radius;
        if constexpr (sizeof(G) == 8) turn_radius = (V * V) / (G(9.81) * M<G>::tan(C[FD_C_LOS_MAX_BANK_RAD]));
        else turn_radius = (V * V) * C[FD_CD_LOS_INV_G_TAN_BANK];
        const G heading_error = wrap_pi<G>(heading_cmd - d.heading);
        const G anticipation = turn_radius * M<G>::abs(heading_error) / deg2rad<G>(90.0);
        if (hd < anticipation && M<G>::abs(heading_error) > deg2rad<G>(20.0)) {
            const G blend = G(1) - (hd / anticipation);
            heading_cmd = wrap_pi<G>(heading_cmd + blend * C[FD_C_LOS_LEAD_ANGLE_RAD] * signv(heading_error));
        }
    } else if (gtype == FD_GUIDANCE_PP) {                                    // :146-178
        const G V = pymax(d.airspeed, G(10));
        G turn_radius;
        if constexpr (sizeof(G) == 8) turn_radius = (V * V) / (G(9.81) * M<G>::tan(C[FD_C_WP_MAX_BANK_RAD]));
        else turn_radius = (V * V) * C[FD_CD_WP_INV_G_TAN_BANK];
        G lookahead = C[FD_C_LOOKAHEAD_TIME] * V;
        const G proximity = C[FD_C_PROXIMITY_SCALE] * turn_radius;
        if (hd < proximity) lookahead *= G(0.6) + G(0.4) * (hd / proximity);
        lookahead = clipv(lookahead, C[FD_C_LOOKAHEAD_MIN], C[FD_C_LOOKAHEAD_MAX]);
        if (hd > G(1)) {
            const G eff = pymin(lookahead, hd);
            // fp64: the reference's own expression.  fp32 glue: atan2 of (e1, e0) scaled by the positive eff / hd IS the
            // line-of-sight angle already in heading_cmd (the carrot lies on the LOS) -- no second atan2, no divisions
            if constexpr (sizeof(G) == 8) heading_cmd = M<G>::atan2((e1 / hd) * eff, (e0 / hd) * eff);
        }
    }
    heading_cmd = wrap_pi<G>(heading_cmd);                                   // :185

    G speed_cmd = (wp[FD_WP_SPEED] != wp[FD_WP_SPEED]) ? d.airspeed : wp[FD_WP_SPEED];   // None -> current airspeed
    G he = M<G>::abs(los - d.heading);                                       // :214-228
    he = M<G>::abs(wrap_pi<G>(he));
    const G td = C[FD_C_TURN_THRESHOLD_DIST];
    if (hd < td && he > C[FD_C_TURN_THRESHOLD_ANGLE_RAD]) {
        const G reduction = C[FD_C_MAX_SPEED_REDUCTION] * (G(1) - hd / td);
        speed_cmd = pymax(speed_cmd * (G(1) - reduction), C[FD_C_MIN_SPEED]);
    }
    return hsa_agent<G>(cfg, st, C, heading_cmd, speed_cmd, wp[FD_WP_ALTITUDE], x, d, dt);
}

// controllers/mission_planner.py:128-184 : 3-D acceptance test on the current waypoint
template <typename G> FD_DEV bool waypoint_reached(const G* C, const G* wp, const G (&x)[FD_NX])
{
    const G e0 = wp[FD_WP_NORTH] - x[0], e1 = wp[FD_WP_EAST] - x[1], e2 = (-wp[FD_WP_ALTITUDE]) - x[2];
    return M<G>::sqrt(e0 * e0 + e1 * e1 + e2 * e2) < C[FD_C_ACCEPTANCE_RADIUS];
}

// ----- rate-control env pieces ---------------------------------------------------------------------------------
template <typename G> struct EnvState {
    G cmd[3], prev_action[4], prev_err[3], sign_changes[3];
    G settle_timer, is_settled, time, sched[4], ep_return;
};

// learned_controllers/envs/rewards.py:75-137 (weights :19-25) + SettlingTimeBonus :193-221
// G = type of the carried env words, A = type the reward is COMPUTED in: A = G = double in the fp64 parity variant; A = float
// in the fp32-evaluation variants (whose state already carries fp32-evaluation error: two fp64 ocml exp, a sqrt and half a
// dozen IEEE divisions per env step bought nothing there).  The settling TIMER stays in G: `timer += dt; timer >= 0.2` is a
// threshold on an accumulated sum (0.02 x 10 is 0.19999999999999998 in fp64, the reference fires one step later) and must
// not move with the arithmetic type: fp64 env words accumulate seconds as the reference does, fp32 env words count steps
// against `settle_steps` (the count at which the fp64 sum first reaches 0.2).
template <typename G, typename A>
FD_DEV A env_reward(EnvState<G>& e, const A (&err)[3], const A (&a)[4], A airspeed, A altitude, A roll, A pitch, double dt,
                    int settle_steps)
{
    // no FMA contraction: the reference (NumPy) has none, and the two builds of the env kernel (register-capped or not) must
    // return bit-identical rewards -- left to the backend, which products get fused depended on the register allocation
#pragma clang fp contract(off)
    const A tracking_error = fdiv(err[0] * err[0] + err[1] * err[1] + err[2] * err[2], A(3));
    const A r_tracking = A(-0.5) * tracking_error;
    const A d0 = a[0] - A(e.prev_action[0]), d1 = a[1] - A(e.prev_action[1]), d2 = a[2] - A(e.prev_action[2]);
    const A r_smooth = A(-0.01) * (d0 * d0 + (d1 * d1 + d2 * d2));
    const A roll_st = M<A>::exp(fdiv(-M<A>::abs(roll), deg2rad<A>(45.0)));
    const A pitch_st = M<A>::exp(fdiv(-M<A>::abs(pitch), deg2rad<A>(30.0)));
    const A as_st = clipv(fdiv(airspeed - A(8), A(12)), A(0), A(1));
    const A alt_st = clipv(fdiv(altitude - A(10), A(90)), A(0), A(1));
    const A r_stab = A(0.3) * fdiv(roll_st + pitch_st + as_st + alt_st, A(4));
    A sc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const A pe = A(e.prev_err[i]);
        const bool flip = (signv(err[i]) != signv(pe)) && (M<A>::abs(pe) > A(0.01));
        sc[i] = A(0.9) * A(e.sign_changes[i]) + (flip ? A(1) : A(0));
        e.sign_changes[i] = G(sc[i]);
        e.prev_err[i] = G(err[i]);
    }
    const A r_osc = A(-0.1) * (sc[0] + (sc[1] + sc[2]));
    A r = r_tracking + r_smooth + r_stab + r_osc + A(1);

    bool settled = true;                                                     // rewards.py:197-209
#pragma unroll
    for (int i = 0; i < 3; ++i) settled = settled && (M<A>::abs(err[i]) < pymax(M<A>::abs(A(e.cmd[i])) * A(0.05), A(0.05)));
    if (settled) {
        if constexpr (sizeof(G) == 8) {
            e.settle_timer += dt;
            if (e.settle_timer >= G(0.2)) { e.is_settled = G(1); r += A(2) * A(dt); }
        } else {                                       // fp32 env words: the timer word counts settled steps (exact in fp32)
            e.settle_timer += G(1);
            if (e.settle_timer >= G(settle_steps)) { e.is_settled = G(1); r += A(2) * A(dt); }
        }
    } else {
        e.settle_timer = G(0);
        e.is_settled = G(0);
    }
    return r;
}

// learned_controllers/envs/rate_env.py:374-408
template <typename G, typename A = G, typename E = G>
FD_DEV void env_observation(const G (&x)[FD_NX], const EnvState<E>& e, A airspeed, A altitude, float (&o)[FD_OBS_DIM])
{
    o[0] = float(x[9]); o[1] = float(x[10]); o[2] = float(x[11]);
    o[3] = float(e.cmd[0]); o[4] = float(e.cmd[1]); o[5] = float(e.cmd[2]);
    o[6] = float(e.cmd[0] - x[9]); o[7] = float(e.cmd[1] - x[10]); o[8] = float(e.cmd[2] - x[11]);
    o[9] = float(airspeed); o[10] = float(altitude);
    o[11] = float(x[6]); o[12] = float(x[7]); o[13] = float(x[8]);
    o[14] = float(e.prev_action[0]); o[15] = float(e.prev_action[1]);
    o[16] = float(e.prev_action[2]); o[17] = float(e.prev_action[3]);
}

// ----- counter-based generator for throughput-mode resets (Philox-4x32-10) ---------------------------------
struct Philox {
    uint32_t c[4], k[2];
    FD_DEV static void round(uint32_t (&c)[4], const uint32_t (&k)[2])
    {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c[0], p1 = uint64_t(0xCD9E8D57u) * c[2];
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k[0], n2 = uint32_t(p0 >> 32) ^ c[3] ^ k[1];
        c[1] = uint32_t(p1); c[3] = uint32_t(p0); c[0] = n0; c[2] = n2;
    }
    FD_DEV void block(uint64_t seed, uint32_t a, uint32_t b, uint32_t cc, uint32_t d, uint32_t (&out)[4])
    {
        uint32_t ctr[4] = { a, b, cc, d };
        uint32_t key[2] = { uint32_t(seed), uint32_t(seed >> 32) };
#pragma unroll
        for (int i = 0; i < 10; ++i) { round(ctr, key); key[0] += 0x9E3779B9u; key[1] += 0xBB67AE85u; }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = ctr[i];
    }
};
FD_DEV float u01(uint32_t r) { return (float(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

}  // namespace fdyn
