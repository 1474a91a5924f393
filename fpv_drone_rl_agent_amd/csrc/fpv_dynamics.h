// fpv_dynamics.h — device-side arithmetic of one hover env-step (docs/SPEC.md), gfx950.
//
// One lane owns one drone and keeps its whole state in registers across the 12 physics
// sub-steps.  Templated on the real type: float is the product path, double exists for
// trajectory-level parity runs.  Nothing here includes or links oracle/.
//
// Reference semantics: simulation/hover.py:224-358 (attitude, obs, reward, bookkeeping);
// airframe constants simulation/drone_models/cf2x/cf2x.yaml:1-19, cf2x.urdf:10-76.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpv {

constexpr int kTile = 64;       // drones per storage tile = one wavefront
constexpr int kGranules = 9;    // 16-byte (fp32) granules of state per drone
constexpr int kStateWords = 36;
constexpr int kObsWords = 20;
constexpr int kSubstepPairs = 6;  // hover.py:24 env_step_ratio; each aviary step = 2 sub-steps
constexpr int kYawGranules = 11, kYawStateWords = 44, kYawObsWords = 12;  // docs/SPEC.md §10 (simulation/yaw.py)

// storage slot (granule*4 + component) -> canonical word of docs/SPEC.md §2
__host__ __device__ constexpr int canon_of_slot(int s) {
    constexpr int map[kStateWords] = {0,  1,  2,  34,  3,  4,  5,  6,  7,  8,  9,  35,
                                      10, 11, 12, 28,  13, 14, 15, 16, 17, 18, 19, 29,
                                      20, 21, 22, 26,  23, 24, 25, 27, 30, 31, 32, 33};
    return map[s];
}

template <typename T>
struct alignas(sizeof(T) * 4) Vec4 {
    T x, y, z, w;
};

// ---------------------------------------------------------------- constants (SPEC §1)
template <typename T>
struct C {
    static constexpr T inv_mass = T(10.0);                  // cf2x.urdf:10
    static constexpr T inv_ixx = T(1.0 / 3e-5);             // cf2x.urdf:12
    static constexpr T inv_izz = T(1.0 / 5e-5);
    static constexpr T izz_minus_iyy = T(5e-5 - 3e-5);
    static constexpr T ixx_minus_izz = T(3e-5 - 5e-5);
    static constexpr T arm = T(0.028);                      // cf2x.urdf:33-76
    static constexpr T kq_over_kf = T(7.94e-12 / 3.16e-10); // cf2x.yaml:3-4
    static constexpr T alpha = T((1.0 / 240.0) / 0.01);     // dt / tau, cf2x.yaml:6
    static constexpr T k_lin = T(0.5 * 1.225 * 1.5 * 3.0e-4);  // cf2x.yaml:9-10
    static constexpr T k_pqr = T(1.0e-4);                   // cf2x.yaml:11
    static constexpr T kp_xy = T(2.0e-2), kp_z = T(4.0e-2); // cf2x.yaml:16
    static constexpr T ki_tc_xy = T(2.5e-7 / 120.0), ki_tc_z = T(1.35e-4 / 120.0);  // ki * T_c
    static constexpr T kd_over_tc_xy = T(5.0e-5 * 120.0);   // kd / T_c (kd_z = 0)
    static constexpr T grav = T(9.81);
    static constexpr T dt = T(1.0 / 240.0);                 // hover.py:23
    static constexpr T half_dt = T(0.5 / 240.0);
    static constexpr T inv_agent_dt = T(40.0);              // hover.py:25
    static constexpr T z_rest = T(0.01);                    // cf2x.urdf:29 half height
    static constexpr T pwm_idle = T(0.05);
    static constexpr T v_clamp = T(50.0), w_clamp = T(100.0);
    static constexpr T panel_x = T(4.98), panel_zc = T(6.0);  // hover.py:116-155
    static constexpr T cam_c = T(0.90630778703664996324);   // cos 25 deg, hover.py:85
    static constexpr T cam_s = T(0.42261826174069943618);   // sin 25 deg
    static constexpr T pi = T(3.14159265358979323846);
    static constexpr T two_pi = T(6.28318530717958647692);
    static constexpr T half_pi = T(1.57079632679489661923);
};

// ---------------------------------------------------------------- math shims
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_(double x) { return 1.0 / x; }
__device__ __forceinline__ float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float ceil_(float x) { return ceilf(x); }
__device__ __forceinline__ double ceil_(double x) { return ceil(x); }
__device__ __forceinline__ float floor_(float x) { return floorf(x); }
__device__ __forceinline__ double floor_(double x) { return floor(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float min_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_(double a, double b) { return fmax(a, b); }
template <typename T>
__device__ __forceinline__ T clip_(T x, T lo, T hi) { return min_(max_(x, lo), hi); }
__device__ __forceinline__ float atan2_(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double atan2_(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float asin_(float x) { return asinf(x); }
__device__ __forceinline__ double asin_(double x) { return asin(x); }
__device__ __forceinline__ void sincos_(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void sincos_(double x, double* s, double* c) { sincos(x, s, c); }

// sin and cos of a half Euler angle: |x| <= pi/2, so a branch-free odd/even polynomial (Taylor to
// x^15 / x^14, truncation < 1e-9) replaces sincosf and its large-argument reduction path.
__device__ __forceinline__ void sincos_half_(float x, float* s, float* c) {
    const float x2 = x * x;
    float ps = -1.0f / 1307674368000.0f;
    ps = ps * x2 + 1.0f / 6227020800.0f;
    ps = ps * x2 - 1.0f / 39916800.0f;
    ps = ps * x2 + 1.0f / 362880.0f;
    ps = ps * x2 - 1.0f / 5040.0f;
    ps = ps * x2 + 1.0f / 120.0f;
    ps = ps * x2 - 1.0f / 6.0f;
    *s = x + x * (x2 * ps);
    float pc = -1.0f / 87178291200.0f;
    pc = pc * x2 + 1.0f / 479001600.0f;
    pc = pc * x2 - 1.0f / 3628800.0f;
    pc = pc * x2 + 1.0f / 40320.0f;
    pc = pc * x2 - 1.0f / 720.0f;
    pc = pc * x2 + 1.0f / 24.0f;
    pc = pc * x2 - 0.5f;
    *c = pc * x2 + 1.0f;
}
__device__ __forceinline__ void sincos_half_(double x, double* s, double* c) { sincos(x, s, c); }

// sin(x)/x and cos(x) as even polynomials in x2 = x^2 for |x| <= 0.37 (|omega| <= 173 rad/s
// after the state clamp => x = |omega| dt / 2 <= 0.361): no sqrt, no division, no branch.
// SPEC §4.5's small-angle branch is the same series truncated; its pi/4 angle clamp cannot
// fire once the per-component clamp of 100 rad/s has been applied.
template <typename T>
__device__ __forceinline__ T sinc_poly(T x2) {
    if constexpr (sizeof(T) == 4) {
        T p = T(1.0 / 362880.0);
        p = p * x2 - T(1.0 / 5040.0);
        p = p * x2 + T(1.0 / 120.0);
        p = p * x2 - T(1.0 / 6.0);
        return p * x2 + T(1.0);
    } else {
        T p = T(1.0 / 1307674368000.0);
        p = p * x2 - T(1.0 / 6227020800.0);
        p = p * x2 + T(1.0 / 39916800.0);
        p = p * x2 - T(1.0 / 362880.0);
        p = p * x2 + T(1.0 / 5040.0);
        p = p * x2 - T(1.0 / 120.0);
        p = p * x2 + T(1.0 / 6.0);
        return T(1.0) - p * x2;
    }
}
template <typename T>
__device__ __forceinline__ T cos_poly(T x2) {
    if constexpr (sizeof(T) == 4) {
        T p = T(-1.0 / 3628800.0);
        p = p * x2 + T(1.0 / 40320.0);
        p = p * x2 - T(1.0 / 720.0);
        p = p * x2 + T(1.0 / 24.0);
        p = p * x2 - T(0.5);
        return p * x2 + T(1.0);
    } else {
        T p = T(1.0 / 20922789888000.0);
        p = p * x2 - T(1.0 / 87178291200.0);
        p = p * x2 + T(1.0 / 479001600.0);
        p = p * x2 - T(1.0 / 3628800.0);
        p = p * x2 + T(1.0 / 40320.0);
        p = p * x2 - T(1.0 / 720.0);
        p = p * x2 + T(1.0 / 24.0);
        p = p * x2 - T(0.5);
        return p * x2 + T(1.0);
    }
}

// ---------------------------------------------------------------- Philox4x32-10 (SPEC §9)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low half (both quarter rate)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// eight N(0,1) draws for (seed, global drone id, sub-step pair): ga for the even sub-step, gb for the odd
__device__ __forceinline__ void normal8(uint64_t seed, uint32_t drone, uint64_t pair, float (&ga)[4], float (&gb)[4]) {
    uint32_t x[4];
    philox4x32_10(drone, (uint32_t)pair, (uint32_t)(pair >> 32), 0x46505631u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), x);
    constexpr float k = 1.0f / 65536.0f;
    // -2 ln u = -2 ln2 log2 u ; v_sin_f32 / v_cos_f32 take their argument in revolutions
    constexpr float m2ln2 = -1.3862943611198906f;
    const float u0 = ((float)(x[0] & 0xFFFFu) + 0.5f) * k, u1 = ((float)(x[0] >> 16) + 0.5f) * k;
    const float u2 = ((float)(x[1] & 0xFFFFu) + 0.5f) * k, u3 = ((float)(x[1] >> 16) + 0.5f) * k;
    const float u4 = ((float)(x[2] & 0xFFFFu) + 0.5f) * k, u5 = ((float)(x[2] >> 16) + 0.5f) * k;
    const float u6 = ((float)(x[3] & 0xFFFFu) + 0.5f) * k, u7 = ((float)(x[3] >> 16) + 0.5f) * k;
    const float r0 = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u0)), r1 = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u2));
    const float r2 = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u4)), r3 = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u6));
    ga[0] = r0 * __builtin_amdgcn_cosf(u1); ga[1] = r0 * __builtin_amdgcn_sinf(u1);
    ga[2] = r1 * __builtin_amdgcn_cosf(u3); ga[3] = r1 * __builtin_amdgcn_sinf(u3);
    gb[0] = r2 * __builtin_amdgcn_cosf(u5); gb[1] = r2 * __builtin_amdgcn_sinf(u5);
    gb[2] = r3 * __builtin_amdgcn_cosf(u7); gb[3] = r3 * __builtin_amdgcn_sinf(u7);
}
__device__ __forceinline__ void normal8(uint64_t seed, uint32_t drone, uint64_t pair, double (&ga)[4], double (&gb)[4]) {
    uint32_t x[4];
    philox4x32_10(drone, (uint32_t)pair, (uint32_t)(pair >> 32), 0x46505631u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), x);
    double u[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[2 * j] = ((double)(x[j] & 0xFFFFu) + 0.5) * (1.0 / 65536.0);
        u[2 * j + 1] = ((double)(x[j] >> 16) + 0.5) * (1.0 / 65536.0);
    }
    double s, c, r;
    r = sqrt(-2.0 * log(u[0])); sincospi(2.0 * u[1], &s, &c); ga[0] = r * c; ga[1] = r * s;
    r = sqrt(-2.0 * log(u[2])); sincospi(2.0 * u[3], &s, &c); ga[2] = r * c; ga[3] = r * s;
    r = sqrt(-2.0 * log(u[4])); sincospi(2.0 * u[5], &s, &c); gb[0] = r * c; gb[1] = r * s;
    r = sqrt(-2.0 * log(u[6])); sincospi(2.0 * u[7], &s, &c); gb[2] = r * c; gb[3] = r * s;
}

// ---------------------------------------------------------------- per-drone registers
template <typename T>
struct Drone {
    T px, py, pz, steps;       // granule 0
    T qx, qy, qz, qw;          // granule 1
    T vx, vy, vz, ep_ret;      // granule 2
    T wx, wy, wz, prev_area;   // granule 3
    T t0, t1, t2, t3;          // granule 4: motor throttle
    T ix, iy, iz, prev_ratio;  // granule 5: PID integrators
    T ex, ey, ez, prev_cx;     // granule 6: PID previous error
    T er, ep, eyaw, prev_cy;   // granule 7: previous Euler angles
    T a0, a1, a2, a3;          // granule 8: previous action
};

template <typename T, int G = kGranules>
__device__ __forceinline__ void load_drone(Drone<T>& d, const Vec4<T>* __restrict__ st, int64_t tile,
                                           int lane) {
    const Vec4<T>* p = st + tile * (G * kTile) + lane;
    Vec4<T> g;
    g = p[0 * kTile]; d.px = g.x; d.py = g.y; d.pz = g.z; d.steps = g.w;
    g = p[1 * kTile]; d.qx = g.x; d.qy = g.y; d.qz = g.z; d.qw = g.w;
    g = p[2 * kTile]; d.vx = g.x; d.vy = g.y; d.vz = g.z; d.ep_ret = g.w;
    g = p[3 * kTile]; d.wx = g.x; d.wy = g.y; d.wz = g.z; d.prev_area = g.w;
    g = p[4 * kTile]; d.t0 = g.x; d.t1 = g.y; d.t2 = g.z; d.t3 = g.w;
    g = p[5 * kTile]; d.ix = g.x; d.iy = g.y; d.iz = g.z; d.prev_ratio = g.w;
    g = p[6 * kTile]; d.ex = g.x; d.ey = g.y; d.ez = g.z; d.prev_cx = g.w;
    g = p[7 * kTile]; d.er = g.x; d.ep = g.y; d.eyaw = g.z; d.prev_cy = g.w;
    g = p[8 * kTile]; d.a0 = g.x; d.a1 = g.y; d.a2 = g.z; d.a3 = g.w;
}

template <typename T, int G = kGranules>
__device__ __forceinline__ void store_drone(const Drone<T>& d, Vec4<T>* __restrict__ st, int64_t tile,
                                            int lane) {
    Vec4<T>* p = st + tile * (G * kTile) + lane;
    p[0 * kTile] = Vec4<T>{d.px, d.py, d.pz, d.steps};
    p[1 * kTile] = Vec4<T>{d.qx, d.qy, d.qz, d.qw};
    p[2 * kTile] = Vec4<T>{d.vx, d.vy, d.vz, d.ep_ret};
    p[3 * kTile] = Vec4<T>{d.wx, d.wy, d.wz, d.prev_area};
    p[4 * kTile] = Vec4<T>{d.t0, d.t1, d.t2, d.t3};
    p[5 * kTile] = Vec4<T>{d.ix, d.iy, d.iz, d.prev_ratio};
    p[6 * kTile] = Vec4<T>{d.ex, d.ey, d.ez, d.prev_cx};
    p[7 * kTile] = Vec4<T>{d.er, d.ep, d.eyaw, d.prev_cy};
    p[8 * kTile] = Vec4<T>{d.a0, d.a1, d.a2, d.a3};
}

// ---------------------------------------------------------------- control tick (SPEC §4.1)
template <typename T>
__device__ __forceinline__ void control_tick(Drone<T>& d, T spx, T spy, T spz, T thrust, T (&pwm)[4]) {
    const T e0 = spx - d.wx, e1 = spy - d.wy, e2 = spz - d.wz;
    d.ix = clip_(d.ix + C<T>::ki_tc_xy * e0, T(-1), T(1));
    d.iy = clip_(d.iy + C<T>::ki_tc_xy * e1, T(-1), T(1));
    d.iz = clip_(d.iz + C<T>::ki_tc_z * e2, T(-1), T(1));
    const T ux = clip_(C<T>::kp_xy * e0 + d.ix + C<T>::kd_over_tc_xy * (e0 - d.ex), T(-1), T(1));
    const T uy = clip_(C<T>::kp_xy * e1 + d.iy + C<T>::kd_over_tc_xy * (e1 - d.ey), T(-1), T(1));
    const T uz = clip_(C<T>::kp_z * e2 + d.iz, T(-1), T(1));
    d.ex = e0; d.ey = e1; d.ez = e2;
    T p0 = thrust - ux - uy - uz;
    T p1 = thrust + ux + uy - uz;
    T p2 = thrust + ux - uy + uz;
    T p3 = thrust - ux + uy + uz;
    const T hi = max_(max_(p0, p1), max_(p2, p3));
    const T sc = hi > T(1) ? rcp_(hi) : T(1);
    p0 *= sc; p1 *= sc; p2 *= sc; p3 *= sc;
    const T lo = min_(min_(p0, p1), min_(p2, p3));
    const T lift = lo < C<T>::pwm_idle ? (C<T>::pwm_idle - lo) * rcp_(T(1) - lo) : T(0);
    pwm[0] = p0 + (T(1) - p0) * lift;
    pwm[1] = p1 + (T(1) - p1) * lift;
    pwm[2] = p2 + (T(1) - p2) * lift;
    pwm[3] = p3 + (T(1) - p3) * lift;
}

// ---------------------------------------------------------------- physics sub-step (SPEC §4.2-6)
// inv_mass / k_lin / k_pqr are compile-time constants for the hover task and per-drone values under the
// yaw task's domain randomisation (SPEC §10)
template <typename T, bool NOISE>
__device__ __forceinline__ void physics_substep(Drone<T>& d, const T (&pwm)[4], T sigma, const T (&g)[4],
                                                T inv_mass = C<T>::inv_mass, T k_lin = C<T>::k_lin,
                                                T k_pqr = C<T>::k_pqr) {
    // motors: first-order lag, multiplicative noise, thrust = thr^2 * f_max (f_max = 1 N)
    d.t0 += C<T>::alpha * (pwm[0] - d.t0);
    d.t1 += C<T>::alpha * (pwm[1] - d.t1);
    d.t2 += C<T>::alpha * (pwm[2] - d.t2);
    d.t3 += C<T>::alpha * (pwm[3] - d.t3);
    if constexpr (NOISE) {
        d.t0 += g[0] * d.t0 * sigma;
        d.t1 += g[1] * d.t1 * sigma;
        d.t2 += g[2] * d.t2 * sigma;
        d.t3 += g[3] * d.t3 * sigma;
    }
    const T f0 = d.t0 * d.t0, f1 = d.t1 * d.t1, f2 = d.t2 * d.t2, f3 = d.t3 * d.t3;
    const T fz = (f0 + f1) + (f2 + f3);
    // props: 0 (+x,-y,-spin) 1 (-x,+y,-spin) 2 (+x,+y,+spin) 3 (-x,-y,+spin)
    T tx = C<T>::arm * ((f1 + f2) - (f0 + f3));
    T ty = C<T>::arm * ((f1 + f3) - (f0 + f2));
    T tz = C<T>::kq_over_kf * ((f2 + f3) - (f0 + f1));
    tx -= k_pqr * abs_(d.wx) * d.wx;
    ty -= k_pqr * abs_(d.wy) * d.wy;
    tz -= k_pqr * abs_(d.wz) * d.wz;

    // rotation matrix (columns = body axes in world)
    const T xx = d.qx * d.qx, yy = d.qy * d.qy, zz = d.qz * d.qz;
    const T xy = d.qx * d.qy, xz = d.qx * d.qz, yz = d.qy * d.qz;
    const T wx_ = d.qw * d.qx, wy_ = d.qw * d.qy, wz_ = d.qw * d.qz;
    const T r00 = T(1) - T(2) * (yy + zz), r01 = T(2) * (xy - wz_), r02 = T(2) * (xz + wy_);
    const T r10 = T(2) * (xy + wz_), r11 = T(1) - T(2) * (xx + zz), r12 = T(2) * (yz - wx_);
    const T r20 = T(2) * (xz - wy_), r21 = T(2) * (yz + wx_), r22 = T(1) - T(2) * (xx + yy);
    // body-frame velocity and drag
    const T vbx = r00 * d.vx + r10 * d.vy + r20 * d.vz;
    const T vby = r01 * d.vx + r11 * d.vy + r21 * d.vz;
    const T vbz = r02 * d.vx + r12 * d.vy + r22 * d.vz;
    const T fbx = -k_lin * abs_(vbx) * vbx;
    const T fby = -k_lin * abs_(vby) * vby;
    const T fbz = fz - k_lin * abs_(vbz) * vbz;
    const T ax = (r00 * fbx + r01 * fby + r02 * fbz) * inv_mass;
    const T ay = (r10 * fbx + r11 * fby + r12 * fbz) * inv_mass;
    const T az = (r20 * fbx + r21 * fby + r22 * fbz) * inv_mass - C<T>::grav;
    // Euler's equation with I = diag(ixx, ixx, izz): omega x I omega = (wy wz (izz-iyy), wz wx (ixx-izz), 0)
    const T gx = d.wy * d.wz * C<T>::izz_minus_iyy;
    const T gy = d.wz * d.wx * C<T>::ixx_minus_izz;
    // semi-implicit Euler + state clamp
    d.vx = clip_(d.vx + ax * C<T>::dt, -C<T>::v_clamp, C<T>::v_clamp);
    d.vy = clip_(d.vy + ay * C<T>::dt, -C<T>::v_clamp, C<T>::v_clamp);
    d.vz = clip_(d.vz + az * C<T>::dt, -C<T>::v_clamp, C<T>::v_clamp);
    d.wx = clip_(d.wx + (tx - gx) * (C<T>::inv_ixx * C<T>::dt), -C<T>::w_clamp, C<T>::w_clamp);
    d.wy = clip_(d.wy + (ty - gy) * (C<T>::inv_ixx * C<T>::dt), -C<T>::w_clamp, C<T>::w_clamp);
    d.wz = clip_(d.wz + tz * (C<T>::inv_izz * C<T>::dt), -C<T>::w_clamp, C<T>::w_clamp);
    d.px += d.vx * C<T>::dt;
    d.py += d.vy * C<T>::dt;
    d.pz += d.vz * C<T>::dt;
    // quaternion exponential of the body rate, then renormalise
    const T x2 = (d.wx * d.wx + d.wy * d.wy + d.wz * d.wz) * (C<T>::half_dt * C<T>::half_dt);
    const T s = sinc_poly(x2) * C<T>::half_dt;
    const T dw = cos_poly(x2);
    const T dx = d.wx * s, dy = d.wy * s, dz = d.wz * s;
    const T nx = d.qw * dx + d.qx * dw + d.qy * dz - d.qz * dy;
    const T ny = d.qw * dy - d.qx * dz + d.qy * dw + d.qz * dx;
    const T nz = d.qw * dz + d.qx * dy - d.qy * dx + d.qz * dw;
    const T nw = d.qw * dw - d.qx * dx - d.qy * dy - d.qz * dz;
    // |n|^2 = 1 + e with |e| ~ 1e-7 (product of two unit quaternions): one Newton step of 1/sqrt
    // around 1 is exact to O(e^2), i.e. identical to the normalisation of SPEC §4.5 at this precision
    const T inv = T(1.5) - T(0.5) * (nx * nx + ny * ny + nz * nz + nw * nw);
    d.qx = nx * inv; d.qy = ny * inv; d.qz = nz * inv; d.qw = nw * inv;
    // ground: stick contact
    if (d.pz < C<T>::z_rest) {
        d.pz = C<T>::z_rest;
        d.vx = T(0); d.vy = T(0);
        d.vz = max_(d.vz, T(0));
        d.wx = T(0); d.wy = T(0);
    }
}

// ---------------------------------------------------------------- attitude (SPEC §5)
template <typename T>
__device__ __forceinline__ void euler_from_quat(T x, T y, T z, T w, T& roll, T& pitch, T& yaw) {
    const T s = T(-2) * (x * z - w * y);
    if (s <= T(-0.99999)) {
        roll = T(0); pitch = -C<T>::half_pi; yaw = T(2) * atan2_(x, -y);
    } else if (s >= T(0.99999)) {
        roll = T(0); pitch = C<T>::half_pi; yaw = T(2) * atan2_(-x, y);
    } else {
        roll = atan2_(T(2) * (y * z + w * x), w * w - x * x - y * y + z * z);
        pitch = asin_(s);
        yaw = atan2_(T(2) * (x * y + w * z), w * w + x * x - y * y - z * z);
    }
}

template <typename T>
__device__ __forceinline__ T wrap_angle(T dlt) {  // hover.py:229
    T m = dlt + C<T>::pi;
    if (m < T(0)) m += C<T>::two_pi;
    else if (m >= C<T>::two_pi) m -= C<T>::two_pi;
    return m - C<T>::pi;
}

// ---------------------------------------------------------------- camera features (SPEC §6)
template <typename T>
__device__ __forceinline__ bool camera_features(const Drone<T>& d, T& cx, T& cy, T& area, T& ratio) {
    const T xx = d.qx * d.qx, yy = d.qy * d.qy, zz = d.qz * d.qz;
    const T xy = d.qx * d.qy, xz = d.qx * d.qz, yz = d.qy * d.qz;
    const T wx_ = d.qw * d.qx, wy_ = d.qw * d.qy, wz_ = d.qw * d.qz;
    const T r00 = T(1) - T(2) * (yy + zz), r01 = T(2) * (xy - wz_), r02 = T(2) * (xz + wy_);
    const T r10 = T(2) * (xy + wz_), r11 = T(1) - T(2) * (xx + zz), r12 = T(2) * (yz - wx_);
    const T r20 = T(2) * (xz - wy_), r21 = T(2) * (yz + wx_), r22 = T(1) - T(2) * (xx + yy);
    // panel centre in the body frame; world y and z unit vectors in the body frame are rows 1, 2 of R
    const T dx = C<T>::panel_x - d.px, dy = -d.py, dz = C<T>::panel_zc - d.pz;
    const T bx = r00 * dx + r10 * dy + r20 * dz;
    const T by = r01 * dx + r11 * dy + r21 * dz;
    const T bz = r02 * dx + r12 * dy + r22 * dz;
    // (depth, up, right) of the centre and of the two half-edge vectors
    const T c = C<T>::cam_c, s = C<T>::cam_s;
    const T dep_c = c * bx + s * bz, up_c = c * bz - s * bx, rt_c = -by;
    const T dep_y = c * r10 + s * r12, up_y = c * r12 - s * r10, rt_y = -r11;
    const T dep_z = c * r20 + s * r22, up_z = c * r22 - s * r20, rt_z = -r21;
    T u[4], v[4];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const T sy = (j == 0 || j == 3) ? T(1) : T(-1);
        const T sz = (j >= 2) ? T(1) : T(-1);
        T dep = dep_c + sy * dep_y + sz * dep_z;
        const T up = up_c + sy * up_y + sz * up_z;
        const T rt = rt_c + sy * rt_y + sz * rt_z;
        if (!(dep > T(0.1))) { ok = false; dep = T(1); }
        const T k = T(64) * rcp_(dep);
        u[j] = T(64) + k * rt;
        v[j] = T(64) - k * up;
        ok = ok && (u[j] >= T(1)) && (u[j] <= T(127)) && (v[j] >= T(1)) && (v[j] <= T(127));
    }
    const T sh = (u[0] * v[1] - u[1] * v[0]) + (u[1] * v[2] - u[2] * v[1]) + (u[2] * v[3] - u[3] * v[2]) +
                 (u[3] * v[0] - u[0] * v[3]);
    const T apx = T(0.5) * abs_(sh);
    ok = ok && (apx >= T(1));
    const T umin = min_(min_(u[0], u[1]), min_(u[2], u[3])), umax = max_(max_(u[0], u[1]), max_(u[2], u[3]));
    const T vmin = min_(min_(v[0], v[1]), min_(v[2], v[3])), vmax = max_(max_(v[0], v[1]), max_(v[2], v[3]));
    const T h = vmax - vmin;
    cx = ok ? T(0.25) * ((u[0] + u[1]) + (u[2] + u[3])) * T(1.0 / 64.0) - T(1) : T(0);
    cy = ok ? T(0.25) * ((v[0] + v[1]) + (v[2] + v[3])) * T(1.0 / 64.0) - T(1) : T(0);
    area = ok ? apx * T(1.0 / 16384.0) : T(0);
    ratio = (ok && h > T(0)) ? (umax - umin) * rcp_(h) : T(0);
    return ok;
}

// ---------------------------------------------------------------- pixel detector (SPEC §6b)
// hover.py:157-222 on a 128 x 128 rendering of the panel, without a renderer: a pixel is red when its centre
// (i + 0.5, j + 0.5) lies in the projected front face.  This header holds the per-row part — the projected corners, the span of
// red pixels of an image row, the border rule ahead of everything else (:180-185); fpv_pixel_tile.h hands each blob to a lane
// group, and fpv_polydp.h works the reference's contour calls (:187-213) from the blob's spans.
//   crossings of row y = j + 0.5: edge k crosses iff (v_k <= y) != (v_{k+1} <= y) — the rule of SPEC §6b as it stands;
//   the four compares feed scalar XORs, a non-crossing edge contributes a NaN that v_min3 / v_max3 ignore.

// a / b to ~1 ulp without the IEEE division sequence (fp32); exact division in fp64
__device__ __forceinline__ float div_(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ double div_(double a, double b) { return a / b; }

// span [fa, fb] (as exact integers in T) of the row of pixel centres y = j + 0.5; false when the row is empty
template <typename T>
__device__ __forceinline__ bool row_span(const T (&u)[4], const T (&v)[4], const T (&slope)[4], T y, T& fa, T& fb) {
    const bool c0 = v[0] <= y, c1 = v[1] <= y, c2 = v[2] <= y, c3 = v[3] <= y;
    const T nan = __builtin_nanf("");
    const T x0 = (c0 != c1) ? u[0] + (y - v[0]) * slope[0] : nan;
    const T x1 = (c1 != c2) ? u[1] + (y - v[1]) * slope[1] : nan;
    const T x2 = (c2 != c3) ? u[2] + (y - v[2]) * slope[2] : nan;
    const T x3 = (c3 != c0) ? u[3] + (y - v[3]) * slope[3] : nan;
    const T xl = min_(min_(x0, x1), min_(x2, x3)), xr = max_(max_(x0, x1), max_(x2, x3));   // NaN operands are ignored
    fa = max_(ceil_(xl - T(0.5)), T(0));
    fb = min_(floor_(xr - T(0.5)), T(127));
    return xl <= xr && fa <= fb;                      // (no crossing: xl = xr = NaN, the first test fails)
}

// projected corners, edge slopes and the rows [jlo, jhi] that can hold pixel centres; false: some corner is behind the
// camera or the quadrilateral lies between two rows of centres (then jlo > jhi as well)
template <typename T>
__device__ __forceinline__ bool pixel_setup(const Drone<T>& d, T (&u)[4], T (&v)[4], T (&slope)[4], T& jlo, T& jhi) {
    const T xx = d.qx * d.qx, yy = d.qy * d.qy, zz = d.qz * d.qz;
    const T xy = d.qx * d.qy, xz = d.qx * d.qz, yz = d.qy * d.qz;
    const T wx_ = d.qw * d.qx, wy_ = d.qw * d.qy, wz_ = d.qw * d.qz;
    const T r00 = T(1) - T(2) * (yy + zz), r01 = T(2) * (xy - wz_), r02 = T(2) * (xz + wy_);
    const T r10 = T(2) * (xy + wz_), r11 = T(1) - T(2) * (xx + zz), r12 = T(2) * (yz - wx_);
    const T r20 = T(2) * (xz - wy_), r21 = T(2) * (yz + wx_), r22 = T(1) - T(2) * (xx + yy);
    const T dx = C<T>::panel_x - d.px, dy = -d.py, dz = C<T>::panel_zc - d.pz;
    const T bx = r00 * dx + r10 * dy + r20 * dz;
    const T by = r01 * dx + r11 * dy + r21 * dz;
    const T bz = r02 * dx + r12 * dy + r22 * dz;
    const T c = C<T>::cam_c, s = C<T>::cam_s;
    const T dep_c = c * bx + s * bz, up_c = c * bz - s * bx, rt_c = -by;
    const T dep_y = c * r10 + s * r12, up_y = c * r12 - s * r10, rt_y = -r11;
    const T dep_z = c * r20 + s * r22, up_z = c * r22 - s * r20, rt_z = -r21;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const T sy = (j == 0 || j == 3) ? T(1) : T(-1);
        const T sz = (j >= 2) ? T(1) : T(-1);
        T dep = dep_c + sy * dep_y + sz * dep_z;
        const T up = up_c + sy * up_y + sz * up_z;
        const T rt = rt_c + sy * rt_y + sz * rt_z;
        if (!(dep > T(0.1))) { ok = false; dep = T(1); }
        const T k = div_(T(64), dep);
        u[j] = T(64) + k * rt;
        v[j] = T(64) - k * up;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = (j + 1) & 3;
        const T dv = v[m] - v[j];
        slope[j] = dv != T(0) ? div_(u[m] - u[j], dv) : T(0);
    }
    const T vmin = min_(min_(v[0], v[1]), min_(v[2], v[3])), vmax = max_(max_(v[0], v[1]), max_(v[2], v[3]));
    const T umin = min_(min_(u[0], u[1]), min_(u[2], u[3])), umax = max_(max_(u[0], u[1]), max_(u[2], u[3]));
    jlo = max_(ceil_(vmin - T(0.5)), T(0));
    jhi = min_(floor_(vmax - T(0.5)), T(127));
    // a quadrilateral wholly to the left or to the right of the image covers no pixel centre: every crossing x lies in
    // [umin, umax], and row_span's [ceil(xl - 1/2), floor(xr - 1/2)] clipped to [0, 127] is empty for xr < 1/2 or xl > 127 1/2
    ok = ok && jlo <= jhi && umax >= T(0.5) && umin <= T(127.5);
    if (!ok) { jlo = T(1); jhi = T(0); }
    return ok;
}

// Border rule ahead of the walk (hover.py:180-185: a red pixel on the image border makes the target invisible).  True when
// one of six rows of the blob — its first and last row and the two rows next to its leftmost and to its rightmost vertex,
// where a convex outline reaches its extreme columns — has a span that starts in column 0, ends in column 127 or lies in
// row 0 / 127.  These are rows of the walk evaluated by the walk's own row_span, so "true" is exactly what the walk would
// find (and a border contact on another row is still found by the walk): skipping the walk then changes no result.
// Nearly half of the blobs that have rows at all touch the border (measured on the synthetic states of bench.py).
template <typename T>
__device__ __forceinline__ bool pixel_touches_border(const T (&u)[4], const T (&v)[4], const T (&slope)[4], T jlo, T jhi) {
    T vl = v[0], ul = u[0], vr = v[0], ur = u[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        if (u[k] < ul) { ul = u[k]; vl = v[k]; }
        if (u[k] > ur) { ur = u[k]; vr = v[k]; }
    }
    const T rl = clip_(floor_(vl - T(0.5)), jlo, jhi), rr = clip_(floor_(vr - T(0.5)), jlo, jhi);
    const T rows[6] = {jlo, jhi, rl, min_(rl + T(1), jhi), rr, min_(rr + T(1), jhi)};
    bool touch = false;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        T a, b;
        const bool ne = row_span(u, v, slope, rows[q] + T(0.5), a, b);
        touch = touch || (ne && (a == T(0) || b == T(127) || rows[q] == T(0) || rows[q] == T(127)));
    }
    return touch && jlo <= jhi;
}

// obs[0:7] of hover.py:224-235: finite-difference Euler rates over agent_dt and the quaternion of the Euler angles
template <typename T>
__device__ __forceinline__ void observe_attitude(T er, T ep, T eyaw, T roll, T pitch, T yaw, T (&o)[7]) {
    o[0] = wrap_angle(roll - er) * C<T>::inv_agent_dt;     // hover.py:228-230
    o[1] = wrap_angle(pitch - ep) * C<T>::inv_agent_dt;
    o[2] = wrap_angle(yaw - eyaw) * C<T>::inv_agent_dt;
    T sr, cr, sp, cp, sy, cy;                              // hover.py:233
    sincos_half_(T(0.5) * roll, &sr, &cr);
    sincos_half_(T(0.5) * pitch, &sp, &cp);
    sincos_half_(T(0.5) * yaw, &sy, &cy);
    o[3] = sr * cp * cy - cr * sp * sy;
    o[4] = cr * sp * cy + sr * cp * sy;
    o[5] = cr * cp * sy - sr * sp * cy;
    o[6] = cr * cp * cy + sr * sp * sy;
}

// obs[7:20] of hover.py:253-267 from the camera features of the caller and the raw action; updates the feature memory.
template <typename T>
__device__ __forceinline__ void observe_features(Drone<T>& d, T a0, T a1, T a2, T a3, T (&o)[kObsWords], bool vis, T cx, T cyy, T area,
                                                 T ratio) {
    o[7] = cx; o[8] = cyy;
    o[9] = d.prev_cx; o[10] = d.prev_cy;
    o[11] = area; o[12] = d.prev_area;
    o[13] = vis ? T(1) : T(0);
    o[14] = ratio; o[15] = d.prev_ratio;
    o[16] = a0; o[17] = a1; o[18] = a2; o[19] = a3;
    d.prev_cx = cx; d.prev_cy = cyy; d.prev_area = area; d.prev_ratio = ratio;  // hover.py:270-272
}

// Observation of hover.py:237-272 into o[20] from the camera features of the caller; updates the feature memory.
template <typename T>
__device__ __forceinline__ void observe_with(Drone<T>& d, T roll, T pitch, T yaw, T a0, T a1, T a2, T a3, T (&o)[kObsWords],
                                             bool vis, T cx, T cyy, T area, T ratio) {
    T att[7];
    observe_attitude(d.er, d.ep, d.eyaw, roll, pitch, yaw, att);
#pragma unroll
    for (int j = 0; j < 7; ++j) o[j] = att[j];
    observe_features(d, a0, a1, a2, a3, o, vis, cx, cyy, area, ratio);
}

// stage a wave's [64][W] observation tile through LDS and write it out row-major (W a multiple of 4)
template <typename T, int W = kObsWords>
__device__ __forceinline__ void store_obs_tile(const T (&o)[W], T* __restrict__ obs, int64_t tile, int lane, int64_t n,
                                               Vec4<T>* lds) {
#pragma unroll
    for (int g = 0; g < W / 4; ++g)
        lds[lane * (W / 4) + g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
    __syncthreads();
    Vec4<T>* out = reinterpret_cast<Vec4<T>*>(obs + tile * (kTile * W));
    const int64_t rows_left = n - tile * kTile;  // rows of this tile that exist
#pragma unroll
    for (int r = 0; r < W / 4; ++r) {
        const int f4 = r * kTile + lane;         // 16-byte granule index inside the tile
        if (f4 / (W / 4) < rows_left) out[f4] = lds[f4];
    }
}

// VecNormalize's per-tile fp64 sums {sum obs[W], sum obs^2[W], sum ret, sum ret^2, 0, 0} (2 W + 4 words) from the
// LDS-staged tile ([64][W] observations followed by 64 returns), by ONE wave: 60 lanes = G row groups x W channels
// (G = 3 for the hover task's 20 words, 5 for the yaw task's 12) walk the tile with fixed trip counts + selects so that
// the LDS reads issue back to back, lanes 60..63 take the returns, and the groups are folded in a fixed order through a
// small LDS exchange (xch: 128 doubles).  Shared by every fused step kernel, so the sums do not depend on which ran.
template <int W>
__device__ __forceinline__ void tile_moment_sums(const float* lds_w, double* xch, int rows, int lane, double* row) {
    constexpr int G = 60 / W, Q = (kTile + G - 1) / G;
    static_assert(G * W == 60, "observation width must divide 60");
    double s1 = 0.0, s2 = 0.0;
    if (lane < G * W) {
        const int g = lane / W, c = lane - g * W;
        // every read is issued unconditionally (the index is clamped into the tile) and masked afterwards: a load under a
        // run-time row test makes hipcc branch around each one and wait for it alone — 22 dependent LDS round trips
        float v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = G * q + g;
            v[q] = lds_w[(r < kTile ? r : 0) * W + c];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const double d = G * q + g < rows ? (double)v[q] : 0.0;
            s1 += d; s2 += d * d;
        }
    } else {
        const int g = lane - G * W;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_w[kTile * W + 4 * q + g];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double d = 4 * q + g < rows ? (double)v[q] : 0.0;
            s1 += d; s2 += d * d;
        }
    }
    xch[2 * lane] = s1;
    xch[2 * lane + 1] = s2;
    // the 64 lanes of this wave exchange through LDS: a wave's LDS operations complete in order, so only the
    // compiler has to be kept from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < W) {
        double a = xch[2 * lane], b = xch[2 * lane + 1];
#pragma unroll
        for (int g = 1; g < G; ++g) { a += xch[2 * (lane + g * W)]; b += xch[2 * (lane + g * W) + 1]; }
        row[lane] = a;
        row[W + lane] = b;
    } else if (lane == 60) {
        row[2 * W] = (xch[120] + xch[122]) + (xch[124] + xch[126]);
        row[2 * W + 1] = (xch[121] + xch[123]) + (xch[125] + xch[127]);
        row[2 * W + 2] = 0.0;
        row[2 * W + 3] = 0.0;
    }
}

}  // namespace fpv
