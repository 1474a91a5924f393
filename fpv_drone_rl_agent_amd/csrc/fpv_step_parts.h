// fpv_step_parts.h — pieces of the hover env step shared by the kernels of fpv_step.hip and fpv_rollstep.hip: the reward /
// termination rules of simulation/hover.py:274-343 and the constant-template reset.
#pragma once
#include "fpv_dynamics.h"
#include "fpv_internal.h"

namespace fpv {

// ---------------------------------------------------------------- reward (SPEC §7)
template <typename T>
__device__ __forceinline__ T reward_and_flags(const Drone<T>& d, T roll, T pitch, bool vis, const T (&o)[kObsWords],
                                              T a0, T a1, T a2, T a3, bool floor_check, uint32_t& bits) {
    T reward = T(-0.1);                                             // hover.py:343
    bits = 0;
    if (d.steps > T(400)) bits |= kInfoTrunc;                       // hover.py:275
    if (sqrt_(d.px * d.px + d.py * d.py + d.pz * d.pz) > T(3.0)) {  // hover.py:278-281
        reward = T(-100);
        bits |= kInfoTerm | kInfoOob;
    }
    if (floor_check && d.steps > T(30) && d.pz < T(0.1)) {          // hover.py:283-290
        reward = T(-100);
        bits |= kInfoTerm | kInfoFloor;
    }
    const T target = vis ? -sqrt_(o[7] * o[7] + o[8] * o[8]) - abs_(o[11] - T(0.013)) - abs_(o[14] - T(1.53))
                         : T(-2);                                   // hover.py:296-320
    reward -= T(0.01) * d.wz * d.wz;                                // hover.py:322-324
    reward += target - sqrt_(roll * roll + pitch * pitch);          // hover.py:326-327
    const T d0 = a0 - d.a0, d1 = a1 - d.a1, d2 = a2 - d.a2, d3 = a3 - d.a3;
    reward -= T(0.2) * sqrt_(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);  // hover.py:329-331
    reward += T(1);                                                 // hover.py:332
    return reward;
}

// The reset template (36 state words in storage-slot order + 20 observation words) is fetched once
// per wave at kernel start, lane j holding word j, so that a finished drone takes it from registers
// (v_readlane) instead of waiting on memory at the end of the kernel.
template <typename T>
__device__ __forceinline__ void drone_from_template(Drone<T>& d, T tv) {
    // prev_action (slots 32..35) is deliberately not taken: hover.py:101 resets only self.action,
    // prev_action survives reset()
    d.px = __shfl(tv, 0); d.py = __shfl(tv, 1); d.pz = __shfl(tv, 2); d.steps = __shfl(tv, 3);
    d.qx = __shfl(tv, 4); d.qy = __shfl(tv, 5); d.qz = __shfl(tv, 6); d.qw = __shfl(tv, 7);
    d.vx = __shfl(tv, 8); d.vy = __shfl(tv, 9); d.vz = __shfl(tv, 10); d.ep_ret = __shfl(tv, 11);
    d.wx = __shfl(tv, 12); d.wy = __shfl(tv, 13); d.wz = __shfl(tv, 14); d.prev_area = __shfl(tv, 15);
    d.t0 = __shfl(tv, 16); d.t1 = __shfl(tv, 17); d.t2 = __shfl(tv, 18); d.t3 = __shfl(tv, 19);
    d.ix = __shfl(tv, 20); d.iy = __shfl(tv, 21); d.iz = __shfl(tv, 22); d.prev_ratio = __shfl(tv, 23);
    d.ex = __shfl(tv, 24); d.ey = __shfl(tv, 25); d.ez = __shfl(tv, 26); d.prev_cx = __shfl(tv, 27);
    d.er = __shfl(tv, 28); d.ep = __shfl(tv, 29); d.eyaw = __shfl(tv, 30); d.prev_cy = __shfl(tv, 31);
}

}  // namespace fpv
