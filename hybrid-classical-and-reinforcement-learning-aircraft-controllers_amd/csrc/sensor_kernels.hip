// sensor_kernels.hip -- the sensor layer between the physics and whoever reads the state (gfx950).
//
// Reference: interfaces/sensor.py:137-243 (NoisySensorInterface): every update adds zero-mean Gaussian noise to position,
// velocity, attitude, body rates (+ a gyro bias), airspeed and altitude, then random-walks the gyro and accelerometer
// biases.  One lane per aircraft, word-major rows, one pass: 14 rows read, 14 + 6 written -- HBM-bound by construction
// (fp64: 272 B per aircraft per update + 160 B when the normals come from the host).
//
// Noise source.  z != NULL: the caller supplies the 20 standard normals of each update in the reference's draw order
//   (FD_SZ_*; NumPy's rng.normal(0, s, k) is 0 + s * standard_normal(k) on the same stream) -- this reproduces the
//   reference bit for bit and is what the parity tests use.  z == NULL: Philox-4x32-10 keyed by (seed, aircraft, *step,
//   block) + Box-Muller in-kernel; *step is a word in device memory the caller bumps on the stream, so graph replays
//   draw fresh noise.
//
// sensor_observe_kernel applies the same model to a rate-control observation row in place (rates + bias, recomputed
// rate errors, airspeed, altitude, attitude): the domain-randomisation hook between the env step and the policy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fdyn.h"
#include "philox.hpp"

namespace {


// 20 standard normals for lane i at this step: 5 Philox blocks, two Box-Muller pairs each
template <typename S>
__device__ __forceinline__ void draw_normals(uint64_t seed, int64_t i, uint32_t step, S (&z)[FD_NSZ])
{
#pragma unroll
    for (int b = 0; b < FD_NSZ / 4; ++b) {
        uint32_t r[4];
        philox4(seed, uint32_t(i), uint32_t(i >> 32), step, 0x60u + uint32_t(b), r);
        const float u0 = (float(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = (float(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
        z[4 * b + 0] = S(ra * __cosf(6.283185307f * u1)); z[4 * b + 1] = S(ra * __sinf(6.283185307f * u1));
        z[4 * b + 2] = S(rb * __cosf(6.283185307f * u3)); z[4 * b + 3] = S(rb * __sinf(6.283185307f * u3));
    }
}

template <typename S>
__global__ void __launch_bounds__(256)
sensor_update_kernel(const S* __restrict__ x /*[FD_NX][n]*/, const S* __restrict__ derived /*[FD_ND][n] or null*/,
                     S* __restrict__ bias /*[FD_NSB][n]*/, const double* __restrict__ cfg /*[FD_NSN]*/,
                     const S* __restrict__ zin /*[FD_NSZ][n] or null*/, uint64_t seed, const uint32_t* __restrict__ step,
                     S* __restrict__ meas /*[FD_NMS][n]*/, int64_t n)
{
#pragma clang fp contract(off)                       // true + (s * z) as the reference rounds it: no FMA
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    S xv[FD_NX];
#pragma unroll
    for (int k = 0; k < FD_NX; ++k) xv[k] = x[int64_t(k) * n + i];
    S airspeed, altitude;
    if (derived) {                                    // AircraftState.airspeed / .altitude as the backend reported them
        airspeed = derived[int64_t(FD_D_AIRSPEED) * n + i];
        altitude = derived[int64_t(FD_D_ALTITUDE) * n + i];
    } else {                                          // simplified_6dof.py:306-309
        airspeed = S(sqrt(double(xv[FD_X_U]) * double(xv[FD_X_U]) + double(xv[FD_X_V]) * double(xv[FD_X_V]) +
                          double(xv[FD_X_W]) * double(xv[FD_X_W])));
        altitude = -xv[FD_X_D];
    }
    if (cfg[FD_SN_ENABLED] == 0.0) {                  // sensor.py:203-205
#pragma unroll
        for (int k = 0; k < FD_NX; ++k) meas[int64_t(k) * n + i] = xv[k];
        meas[int64_t(FD_MS_AIRSPEED) * n + i] = airspeed;
        meas[int64_t(FD_MS_ALTITUDE) * n + i] = altitude;
        return;
    }
    S z[FD_NSZ];
    if (zin) {
#pragma unroll
        for (int k = 0; k < FD_NSZ; ++k) z[k] = zin[int64_t(k) * n + i];
    } else {
        draw_normals<S>(seed, i, step ? *step : 0u, z);
    }
    const S s_pos = S(cfg[FD_SN_GPS_POS]), s_vel = S(cfg[FD_SN_GPS_VEL]), s_att = S(cfg[FD_SN_ATTITUDE]);
    const S s_gyro = S(cfg[FD_SN_GYRO]), s_as = S(cfg[FD_SN_AIRSPEED]), s_alt = S(cfg[FD_SN_ALTITUDE]);
    const S w_g = S(cfg[FD_SN_GYRO_BIAS_WALK]), w_a = S(cfg[FD_SN_ACCEL_BIAS_WALK]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const S bg = bias[int64_t(k) * n + i], ba = bias[int64_t(3 + k) * n + i];
        meas[int64_t(FD_X_N + k) * n + i] = xv[FD_X_N + k] + s_pos * z[FD_SZ_POS + k];               // :208-210
        meas[int64_t(FD_X_U + k) * n + i] = xv[FD_X_U + k] + s_vel * z[FD_SZ_VEL + k];               // :211-213
        meas[int64_t(FD_X_ROLL + k) * n + i] = xv[FD_X_ROLL + k] + s_att * z[FD_SZ_ATT + k];         // :214-216
        meas[int64_t(FD_X_P + k) * n + i] = (xv[FD_X_P + k] + s_gyro * z[FD_SZ_GYRO + k]) + bg;      // :217-221
        bias[int64_t(k) * n + i] = bg + w_g * z[FD_SZ_GYRO_BIAS + k];                                // :230
        bias[int64_t(3 + k) * n + i] = ba + w_a * z[FD_SZ_ACCEL_BIAS + k];                           // :231
    }
    meas[int64_t(FD_MS_AIRSPEED) * n + i] = airspeed + s_as * z[FD_SZ_AIRSPEED];                     // :222-224
    meas[int64_t(FD_MS_ALTITUDE) * n + i] = altitude + s_alt * z[FD_SZ_ALTITUDE];                    // :225-227
}

// In-place noise on rate-control observations [n][18] (layout rate_env.py:374-408), fp32.
__global__ void __launch_bounds__(256)
sensor_observe_kernel(float* __restrict__ obs /*[n][18]*/, float* __restrict__ gyro_bias /*[3][n]*/,
                      const uint8_t* __restrict__ reset_mask /*[n] or null*/, const double* __restrict__ cfg,
                      const float* __restrict__ zin /*[FD_NSZ][n] or null*/, uint64_t seed, const uint32_t* __restrict__ step,
                      int64_t n)
{
#pragma clang fp contract(off)
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n || cfg[FD_SN_ENABLED] == 0.0) return;
    float z[FD_NSZ];
    if (zin) {
#pragma unroll
        for (int k = 0; k < FD_NSZ; ++k) z[k] = zin[int64_t(k) * n + i];
    } else {
        draw_normals<float>(seed, i, step ? *step : 0u, z);
    }
    const bool fresh = reset_mask && reset_mask[i];                  // first observation of a new episode: sensor.reset()
    float* o = obs + i * FD_OBS_DIM;
    const float s_att = float(cfg[FD_SN_ATTITUDE]), s_gyro = float(cfg[FD_SN_GYRO]);
    const float s_as = float(cfg[FD_SN_AIRSPEED]), s_alt = float(cfg[FD_SN_ALTITUDE]), w_g = float(cfg[FD_SN_GYRO_BIAS_WALK]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float bg = fresh ? 0.0f : gyro_bias[int64_t(k) * n + i];
        const float rate = (o[k] + s_gyro * z[FD_SZ_GYRO + k]) + bg;
        o[k] = rate;
        o[6 + k] = o[3 + k] - rate;                                   // the error the controller can actually see
        o[11 + k] = o[11 + k] + s_att * z[FD_SZ_ATT + k];
        gyro_bias[int64_t(k) * n + i] = bg + w_g * z[FD_SZ_GYRO_BIAS + k];
    }
    o[9] = o[9] + s_as * z[FD_SZ_AIRSPEED];
    o[10] = o[10] + s_alt * z[FD_SZ_ALTITUDE];
}

template <typename S>
int launch_update(const S* x, const S* derived, S* bias, const double* cfg, const S* z, uint64_t seed, const uint32_t* step,
                  S* meas, int64_t n, void* stream)
{
    if (n < 0) return FDYN_ERR_BAD_SIZE;
    if (n == 0) return FDYN_OK;
    if (!x || !bias || !cfg || !meas) return FDYN_ERR_NULL;
    hipLaunchKernelGGL(sensor_update_kernel<S>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, derived,
                       bias, cfg, z, seed, step, meas, n);
    return int(hipGetLastError());
}

}  // namespace

extern "C" {

int fdyn_sensor_update_f64(const double* x, const double* derived, double* bias, const double* noise_cfg, const double* z,
                           uint64_t seed, const uint32_t* step, double* meas, int64_t n, void* stream)
{
    return launch_update<double>(x, derived, bias, noise_cfg, z, seed, step, meas, n, stream);
}

int fdyn_sensor_update_f32(const float* x, const float* derived, float* bias, const double* noise_cfg, const float* z,
                           uint64_t seed, const uint32_t* step, float* meas, int64_t n, void* stream)
{
    return launch_update<float>(x, derived, bias, noise_cfg, z, seed, step, meas, n, stream);
}

int fdyn_sensor_observe(float* obs, float* gyro_bias, const uint8_t* reset_mask, const double* noise_cfg, const float* z,
                        uint64_t seed, const uint32_t* step, int64_t n, void* stream)
{
    if (n < 0) return FDYN_ERR_BAD_SIZE;
    if (n == 0) return FDYN_OK;
    if (!obs || !gyro_bias || !noise_cfg) return FDYN_ERR_NULL;
    hipLaunchKernelGGL(sensor_observe_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs,
                       gyro_bias, reset_mask, noise_cfg, z, seed, step, n);
    return int(hipGetLastError());
}

}  // extern "C"
