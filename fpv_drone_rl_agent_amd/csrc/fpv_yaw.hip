// fpv_yaw.hip — yaw-setpoint task (simulation/yaw.py skeleton + docs/SPEC.md §10), BASELINE cfg 5.
//
// Same machinery as the hover step (one lane per drone, 64-drone tiles of 16-byte granules, LDS-staged
// observation rows, per-tile done ledger), with: a 1-D action, ONE aviary step (2 physics sub-steps) per
// agent step, a 12-word observation, per-drone domain-randomised mass / drag (two extra granules of
// state: action history and airframe parameters), and a reset that draws the initial heading and the
// airframe from Philox inside the kernel.
#include "fpv_dynamics.h"
#include "fpv_internal.h"
#include "fpv_yaw_parts.h"

namespace fpv {

// FUSED / TPB as in the hover step (fpv_step.hip): VecNormalize's done mask, discounted return and moment sums in the same
// pass, TPB tiles per workgroup with one partial row (2 x 12 + 4 words) per workgroup.
template <typename T, bool NOISE, bool FUSED = false, int TPB = 1>
__global__ __launch_bounds__(kTile * TPB) void yaw_step_kernel(StepArgs<T> p) {
    constexpr int kLdsPerTile = kTile * kYawObsWords / 4 + (FUSED ? kTile / 4 + kTile : 0);
    constexpr int kRow = 2 * kYawObsWords + 4;
    __shared__ Vec4<T> lds_all[TPB * kLdsPerTile];
    __shared__ double bsum[TPB > 1 ? TPB * kRow : 1];
    const int lane = threadIdx.x & (kTile - 1), wv = threadIdx.x / kTile;
    Vec4<T>* lds = lds_all + wv * kLdsPerTile;
    const int64_t tile = (int64_t)blockIdx.x * TPB + wv;
    const int64_t i = tile * kTile + lane;
    const bool live = i < p.n;
    T ret_in = T(0);
    if constexpr (FUSED) {
        if (p.ret && live) ret_in = p.ret[i];
    }

    Drone<T> d;
    load_drone<T, kYawGranules>(d, p.state, tile, lane);
    Vec4<T>* ext = p.state + tile * (kYawGranules * kTile) + lane;
    Vec4<T> g9 = ext[9 * kTile], g10 = ext[10 * kTile];
    YawExtra<T> e{g9.x, g9.y, g9.z, g9.w, g10.x, g10.y, g10.z, g10.w};
    const T a = live ? p.actions[i] : T(0);
    const uint64_t agent_tick = p.tick[tile];
    const uint32_t led_fill = p.led_count[tile];
    const uint32_t gid = (uint32_t)(p.id_offset + (uint64_t)i);

    const T prev_a = e.h3;
    e.h0 = e.h1; e.h1 = e.h2; e.h2 = e.h3; e.h3 = a;                   // yaw.py:107
    // yaw.py:110-121: full action (0, 0, a, -1) -> setpoint (0, 0, -30 a, 0)
    T pwm[4];
    control_tick(d, T(0), T(0), a * T(-30), T(0), pwm);
    T ga[4] = {}, gb[4] = {};
    if constexpr (NOISE) normal8(p.seed, gid, agent_tick, ga, gb);
    const T inv_mass = rcp_(e.mass), k_lin = T(0.5 * 1.225 * 3.0e-4) * e.cd;
    physics_substep<T, NOISE>(d, pwm, p.sigma, ga, inv_mass, k_lin, e.kpqr);   // yaw.py:126-127: one aviary step
    physics_substep<T, NOISE>(d, pwm, p.sigma, gb, inv_mass, k_lin, e.kpqr);

    T roll, pitch, yaw;
    euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
    uint32_t bits = 0;
    if (d.steps >= T(400)) bits |= kInfoTrunc;                          // yaw.py:138
    if (sqrt_(d.px * d.px + d.py * d.py + d.pz * d.pz) > T(3.0)) bits |= kInfoTerm | kInfoOob;  // yaw.py:141-143
    d.steps += T(1);                                                    // yaw.py:145
    const T av0 = clip_(wrap_angle(roll - d.er) * T(4), T(-1), T(1));   // * 120 / 30
    const T av1 = clip_(wrap_angle(pitch - d.ep) * T(4), T(-1), T(1));
    const T av2 = clip_(wrap_angle(yaw - d.eyaw) * T(4), T(-1), T(1));
    T o[kYawObsWords];
    const bool vis = yaw_observe(d, e, roll, pitch, yaw, av0, av1, av2, o);
    const T target = vis ? -sqrt_(o[3] * o[3] + o[4] * o[4]) : T(-2);
    const T reward = T(1) + target - T(0.01) * d.wz * d.wz - T(0.2) * abs_(a - prev_a);
    d.er = roll; d.ep = pitch; d.eyaw = yaw;
    d.ep_ret += reward;

    const bool done = live && bits != 0;
    const unsigned long long m = __ballot(done);
    if (m != 0ull) {
        if (done) {
            const int64_t row = (int64_t)led_fill + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (row < p.led_rows) {
                const int64_t slot = tile * p.led_rows + row;
                p.led_idx[slot] = (int32_t)i;
                p.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                p.led_tag[slot] = (uint32_t)agent_tick;
                p.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(p.led_obs + slot * kYawObsWords);
#pragma unroll
                for (int g = 0; g < kYawObsWords / 4; ++g) lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            yaw_reset(d, e, warm_throttle<T>(), p.seed, gid, agent_tick, p.domain_rand != 0, o);
        }
        if (lane == 0) p.led_count[tile] = led_fill + (uint32_t)__popcll(m);
    }

    store_drone<T, kYawGranules>(d, p.state, tile, lane);
    ext[9 * kTile] = Vec4<T>{e.h0, e.h1, e.h2, e.h3};
    ext[10 * kTile] = Vec4<T>{e.mass, e.cd, e.kpqr, e.spare};
    if (lane == 0) p.tick[tile] = agent_tick + 1ull;
    if (live) {
        p.reward[i] = reward;
        p.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        p.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
    if constexpr (FUSED) {
        T* lds_w = reinterpret_cast<T*>(lds);
        if (live && p.done) p.done[i] = done ? 1 : 0;
        T ret = T(0);
        if (p.ret) {
            ret = live ? ret_in * (T)p.gamma + reward : T(0);
            if (live) p.ret[i] = done ? T(0) : ret;
        }
        lds_w[kTile * kYawObsWords + lane] = ret;
        store_obs_tile<T, kYawObsWords>(o, p.obs, tile, lane, p.n, lds);
        if (p.partials) {
            if constexpr (sizeof(T) == 4) {
                const int rows = (int)((p.n - tile * kTile) < kTile ? (p.n - tile * kTile) : kTile);
                tile_moment_sums<kYawObsWords>(reinterpret_cast<const float*>(lds),
                                               reinterpret_cast<double*>(lds_w + kTile * kYawObsWords + kTile), rows, lane,
                                               TPB > 1 ? bsum + wv * kRow : p.partials + tile * kRow);
                if constexpr (TPB == 4) {
                    __syncthreads();
                    if (wv == 0 && lane < kRow) {
                        const double v = (bsum[lane] + bsum[kRow + lane]) + (bsum[2 * kRow + lane] + bsum[3 * kRow + lane]);
                        double* row = p.partials + (int64_t)blockIdx.x * kRow + lane;
#ifdef FPV_EXPERIMENTS
                        if (p.mg_ticket) row_store_sc1(row, v);      // write-through: read by the last workgroup of this launch
                        else
#endif
                        *row = v;
                    }
#ifdef FPV_EXPERIMENTS                                                  // (merge_in_step: measured slower)
                    if (p.mg_ticket) {                                  // uniform
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        step_merge_tail(p, p.partials, kYawObsWords, bsum);
                    }
#endif
                }
            }
        }
    } else {
        store_obs_tile<T, kYawObsWords>(o, p.obs, tile, lane, p.n, lds);
    }
}

template <typename T>
__global__ __launch_bounds__(kTile) void yaw_reset_kernel(Vec4<T>* state, const uint64_t* tick, const uint8_t* mask, T* obs,
                                                          int64_t n, uint64_t seed, uint64_t id_offset, int domain_rand,
                                                          int all_lanes) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t i = tile * kTile + lane;
    if (!all_lanes && i >= n) return;
    if (mask && i < n && !mask[i]) return;
    Drone<T> d;
    YawExtra<T> e;
    T o[kYawObsWords];
    yaw_reset(d, e, warm_throttle<T>(), seed, (uint32_t)(id_offset + (uint64_t)i), tick[tile], domain_rand != 0, o);
    store_drone<T, kYawGranules>(d, state, tile, lane);
    Vec4<T>* ext = state + tile * (kYawGranules * kTile) + lane;
    ext[9 * kTile] = Vec4<T>{e.h0, e.h1, e.h2, e.h3};
    ext[10 * kTile] = Vec4<T>{e.mass, e.cd, e.kpqr, e.spare};
    if (obs && i < n) {
        T* row = obs + i * kYawObsWords;
#pragma unroll
        for (int j = 0; j < kYawObsWords; ++j) row[j] = o[j];
    }
}

// canonical [n][44] <-> tiled granule storage (first 36 words as the hover map, the rest in place)
template <typename T, bool TO_CANON>
__global__ void yaw_convert_kernel(Vec4<T>* state, T* canon, int64_t n) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * kYawStateWords) return;
    const int64_t i = q / kYawStateWords;
    const int slot = (int)(q % kYawStateWords);
    T* word = reinterpret_cast<T*>(state + (i / kTile) * (kYawGranules * kTile) + (slot / 4) * kTile + (i % kTile)) + (slot % 4);
    T* c = canon + i * kYawStateWords + (slot < kStateWords ? canon_of_slot(slot) : slot);
    if (TO_CANON) *c = *word;
    else *word = *c;
}

// ---------------------------------------------------------------- launchers
template <typename T>
hipError_t launch_yaw_step(const StepArgs<T>& a_in, bool noise, hipStream_t s, bool* merged) {
    StepArgs<T> a = a_in;
    const int64_t tiles = (a.n + kTile - 1) / kTile;
    const bool fused = a.done || a.ret || a.partials;
    const bool can_merge = sizeof(T) == 4 && fused && a.partials && a.mg_ticket && tiles % 4 == 0 && tiles / 4 <= 4096;
#ifndef FPV_EXPERIMENTS
    a.mg_ticket = nullptr;                             // the in-launch merge is an FPV_EXPERIMENTS build's
    if (merged) *merged = false;
    (void)can_merge;
#else
    if (!can_merge) a.mg_ticket = nullptr;             // (the in-launch merge lives in the four-tiles-per-workgroup variant)
    if (merged) *merged = can_merge;
#endif
    if constexpr (sizeof(T) == 4) {
        if (fused) {
            if (tiles % 4 == 0) {               // four tiles per workgroup, one partial row per workgroup (yaw_partial_rows)
                if (noise) hipLaunchKernelGGL((yaw_step_kernel<T, true, true, 4>), dim3((unsigned)(tiles / 4)), dim3(4 * kTile), 0, s, a);
                else hipLaunchKernelGGL((yaw_step_kernel<T, false, true, 4>), dim3((unsigned)(tiles / 4)), dim3(4 * kTile), 0, s, a);
            } else {
                if (noise) hipLaunchKernelGGL((yaw_step_kernel<T, true, true, 1>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
                else hipLaunchKernelGGL((yaw_step_kernel<T, false, true, 1>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
            }
            return hipGetLastError();
        }
    }
    if (noise) hipLaunchKernelGGL((yaw_step_kernel<T, true>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
    else hipLaunchKernelGGL((yaw_step_kernel<T, false>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
    return hipGetLastError();
}

int64_t yaw_partial_rows(int64_t n) {
    const int64_t tiles = (n + kTile - 1) / kTile;
    return tiles % 4 == 0 ? tiles / 4 : tiles;
}
template hipError_t launch_yaw_step<float>(const StepArgs<float>&, bool, hipStream_t, bool*);
template hipError_t launch_yaw_step<double>(const StepArgs<double>&, bool, hipStream_t, bool*);

template <typename T>
hipError_t launch_yaw_reset(void* state, const uint64_t* tick, const uint8_t* mask, void* obs, int64_t n, uint64_t seed,
                            uint64_t id_offset, int domain_rand, int all_lanes, hipStream_t s) {
    const int64_t tiles = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(yaw_reset_kernel<T>, dim3((unsigned)tiles), dim3(kTile), 0, s, (Vec4<T>*)state, tick, mask, (T*)obs, n,
                       seed, id_offset, domain_rand, all_lanes);
    return hipGetLastError();
}
template hipError_t launch_yaw_reset<float>(void*, const uint64_t*, const uint8_t*, void*, int64_t, uint64_t, uint64_t, int,
                                            int, hipStream_t);
template hipError_t launch_yaw_reset<double>(void*, const uint64_t*, const uint8_t*, void*, int64_t, uint64_t, uint64_t, int,
                                             int, hipStream_t);

template <typename T>
hipError_t launch_yaw_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s) {
    const int64_t q = n * kYawStateWords;
    const dim3 grid((unsigned)((q + 255) / 256)), block(256);
    if (to_canon) hipLaunchKernelGGL((yaw_convert_kernel<T, true>), grid, block, 0, s, (Vec4<T>*)state, (T*)canon, n);
    else hipLaunchKernelGGL((yaw_convert_kernel<T, false>), grid, block, 0, s, (Vec4<T>*)state, (T*)canon, n);
    return hipGetLastError();
}
template hipError_t launch_yaw_convert<float>(void*, void*, int64_t, bool, hipStream_t);
template hipError_t launch_yaw_convert<double>(void*, void*, int64_t, bool, hipStream_t);

}  // namespace fpv
