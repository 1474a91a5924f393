// fpv_yaw_parts.h — pieces of the yaw-setpoint task (simulation/yaw.py skeleton + docs/SPEC.md §10) shared by the kernels of
// fpv_yaw.hip and fpv_rollstep.hip: the two extra state granules, the observation of yaw.py:57-74 and the in-kernel reset.
#pragma once
#include "fpv_dynamics.h"
#include "fpv_internal.h"

namespace fpv {

template <typename T>
struct YawExtra {
    T h0, h1, h2, h3;        // granule 9: last four yaw actions, oldest first (yaw.py:53-55)
    T mass, cd, kpqr, spare; // granule 10: airframe of this episode
};

template <typename T>
__device__ __forceinline__ void yaw_uniforms(uint64_t seed, uint32_t gid, uint64_t tick, T (&u)[4]) {
    uint32_t x[4];
    philox4x32_10(gid, (uint32_t)tick, (uint32_t)(tick >> 32), 0x59415731u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = ((T)(x[j] >> 8) + T(0.5)) * T(1.0 / 16777216.0);
}

// observation of yaw.py:57-74 (SPEC §10.4); returns the target visibility
template <typename T>
__device__ __forceinline__ bool yaw_observe(const Drone<T>& d, const YawExtra<T>& e, T roll, T pitch, T yaw, T av0, T av1,
                                            T av2, T (&o)[kYawObsWords]) {
    constexpr T inv_pi = T(0.31830988618379067154);
    T cx, cy, area, ratio;
    const bool vis = camera_features(d, cx, cy, area, ratio);
    o[0] = roll * inv_pi; o[1] = pitch * inv_pi; o[2] = yaw * inv_pi;   // yaw.py:59
    o[3] = cx; o[4] = cy;                                               // yaw.py:62-63
    o[5] = av0; o[6] = av1; o[7] = av2;                                 // yaw.py:66
    o[8] = e.h0; o[9] = e.h1; o[10] = e.h2; o[11] = e.h3;               // yaw.py:69
    return vis;
}

// SPEC §10 reset: resting on the floor, random heading, (optionally) a new airframe
template <typename T>
__device__ __forceinline__ void yaw_reset(Drone<T>& d, YawExtra<T>& e, T thr_warm, uint64_t seed, uint32_t gid,
                                          uint64_t tick, bool domain_rand, T (&o)[kYawObsWords]) {
    T u[4];
    yaw_uniforms(seed, gid, tick, u);
    const T psi = C<T>::pi * (T(2) * u[0] - T(1));                      // yaw.py:79
    T s, c;
    sincos_(T(0.5) * psi, &s, &c);
    d = Drone<T>{};
    d.pz = C<T>::z_rest;
    d.qz = s; d.qw = c;
    d.t0 = d.t1 = d.t2 = d.t3 = thr_warm;
    d.eyaw = psi;
    e.h0 = e.h1 = e.h2 = e.h3 = T(0);
    e.mass = domain_rand ? T(0.08) + T(0.04) * u[1] : T(0.1);
    e.cd = domain_rand ? T(1) + u[2] : T(1.5);
    e.kpqr = domain_rand ? (T(0.5) + T(1.5) * u[3]) * T(1.0e-4) : T(1.0e-4);
    e.spare = T(0);
    yaw_observe(d, e, T(0), T(0), psi, T(0), T(0), T(0), o);
}

template <typename T>
__device__ __forceinline__ T warm_throttle() {
    T thr = T(0);
    for (int k = 0; k < 20; ++k) thr += C<T>::alpha * (C<T>::pwm_idle - thr);  // hover.py:109-110, noise-free
    return thr;
}

}  // namespace fpv
