// fpv_step.hip — the fused env-step kernel and its small companions (gfx950).
//
// K1 step_kernel: one lane per drone, one 64-lane workgroup per storage tile.
//   HBM traffic per drone and agent step: 9 granule loads + 9 granule stores (16 B per lane,
//   1 KiB contiguous per wave instruction), one 16-B action load, 80 B of observation staged
//   through LDS so that the [N][20] rows leave as five 1-KiB contiguous stores per wave, 4 B
//   reward, 2 flag bytes.  All 12 physics sub-steps run in registers.
//   Episode ends are compacted into the tile's own region of the done ledger with one wave
//   ballot (no atomic); the reset state is a constant template (docs/SPEC.md §8), so a
//   finished drone costs its wave a few selects, not a divergent warm-up loop.
//
// Reference path replaced: QuadXHoverEnv.step()/reset() simulation/hover.py:72-114,334-358
// under the SB3 VecEnv auto-reset protocol (simulation/train_hover.py:42).
#include "fpv_dynamics.h"
#include "fpv_internal.h"
#include "fpv_pixel_tile.h"
#include "fpv_step_parts.h"

namespace fpv {

// Observation of hover.py:237-272 with the handle's detector.  The pixel detector's rows and contours are worked by the tile's
// wave together (fpv_pixel_tile.h: `pix`, LDS of this wave) — called by all 64 lanes, for fp32 and fp64 handles alike.
template <typename T>
__device__ __forceinline__ bool observe_tile(Drone<T>& d, T roll, T pitch, T yaw, T a0, T a1, T a2, T a3, T (&o)[kObsWords], int detector,
                                             float* pix, int lane) {
    T cx, cy, area, ratio;
    const bool vis = detector ? camera_features_pixel_tile(d, pix, lane, cx, cy, area, ratio) : camera_features(d, cx, cy, area, ratio);
    observe_with(d, roll, pitch, yaw, a0, a1, a2, a3, o, vis, cx, cy, area, ratio);
    return vis;
}

// ---------------------------------------------------------------- K1
// TPB tiles (waves) per workgroup.  The waves are independent except at the very end of the fused variant, where a
// TPB = 4 workgroup adds its four tiles' moment sums and writes ONE partial row: the table that the statistics merge
// (fpv_moments_merge, or the policy kernel's prologue) has to read shrinks fourfold.
template <typename T, bool NOISE, bool FUSED, int TPB = 1>
__global__ __launch_bounds__(kTile * TPB) void step_kernel(StepArgs<T> p) {
    // per wave: observation tile; fused variant: + 64 returns + a 64 x 2 fp64 exchange area (fp32 handles only).  The pixel
    // detector's tile area is used before the observation exists: the two share the wave's LDS.
    constexpr int kLdsPerTile = kTile * kObsWords / 4 + (FUSED ? kTile / 4 + kTile : 0);
    constexpr int kObsBytes = kLdsPerTile * (int)sizeof(Vec4<T>), kPixBytes = OneWavePixT<T>::kWords * 4;
    constexpr int kWaveBytes = kObsBytes > kPixBytes ? kObsBytes : kPixBytes;
    __shared__ __attribute__((aligned(32))) unsigned char lds_raw[TPB * kWaveBytes];
    __shared__ double bsum[TPB > 1 ? TPB * kPartialWords : 1];
    const int lane = threadIdx.x & (kTile - 1), wv = threadIdx.x / kTile;
    Vec4<T>* lds = reinterpret_cast<Vec4<T>*>(lds_raw + wv * kWaveBytes);
    float* pix = reinterpret_cast<float*>(lds_raw + wv * kWaveBytes);
    const int64_t tile = (int64_t)blockIdx.x * TPB + wv;
    const int64_t i = tile * kTile + lane;
    const bool live = i < p.n;

    Drone<T> d;
    load_drone(d, p.state, tile, lane);
    const Vec4<T> act = live ? reinterpret_cast<const Vec4<T>*>(p.actions)[i] : Vec4<T>{T(0), T(0), T(0), T(-1)};
    // per-tile copy of the agent tick: read and advanced by its own wave only, so the step stays
    // one kernel and is graph-replayable without host-side argument updates
    const uint64_t agent_tick = p.tick[tile];
    const uint32_t led_fill = p.led_count[tile];
    const T tmpl_lane = p.tmpl[lane < kStateWords + kObsWords ? lane : 0];
    T ret_in = T(0);   // VecNormalize's discounted return: requested with the state so its latency hides under the physics
    if constexpr (FUSED) {
        if (p.ret && live) ret_in = p.ret[i];
    }
    const uint32_t gid = (uint32_t)(p.id_offset + (uint64_t)i);

    // hover.py:337-341 action scaling; the mode-0 thrust setpoint is clipped to [0,1] (SPEC §4.1)
    const T spx = act.x * T(30), spy = act.y * T(30), spz = act.z * T(-30);
    const T thrust = clip_((act.w + T(1)) * T(0.5), T(0), T(1));

    // hover.py:346-349: 6 aviary steps = 6 x (control tick + 2 physics sub-steps)
    T pwm[4];
#pragma unroll 1
    for (int k = 0; k < kSubstepPairs; ++k) {
        control_tick(d, spx, spy, spz, thrust, pwm);
        T ga[4] = {}, gb[4] = {};
        if constexpr (NOISE) normal8(p.seed, gid, agent_tick * (uint64_t)kSubstepPairs + (uint64_t)k, ga, gb);
        physics_substep<T, NOISE>(d, pwm, p.sigma, ga);
        physics_substep<T, NOISE>(d, pwm, p.sigma, gb);
    }

    // hover.py:351-357
    T roll, pitch, yaw;
    euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
    T o[kObsWords];
    const bool vis = observe_tile(d, roll, pitch, yaw, act.x, act.y, act.z, act.w, o, p.detector, pix, lane);
    uint32_t bits;
    const T reward = reward_and_flags(d, roll, pitch, vis, o, act.x, act.y, act.z, act.w, p.floor_check != 0, bits);
    d.er = roll; d.ep = pitch; d.eyaw = yaw;
    d.steps += T(1);
    d.a0 = act.x; d.a1 = act.y; d.a2 = act.z; d.a3 = act.w;
    d.ep_ret += reward;

    // episode ends -> this tile's region of the done ledger (wave-ballot compaction, no atomics),
    // then constant-template reset
    const bool done = live && (bits & (kInfoTerm | kInfoTrunc)) != 0;
    const unsigned long long m = __ballot(done);
    if (m != 0ull) {
        Drone<T> r = d;
        drone_from_template(r, tmpl_lane);
        T ro[kObsWords];
#pragma unroll
        for (int j = 0; j < kObsWords; ++j) ro[j] = __shfl(tmpl_lane, kStateWords + j);
        if (done) {
            const int64_t row = (int64_t)led_fill + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (row < p.led_rows) {
                const int64_t slot = tile * p.led_rows + row;
                p.led_idx[slot] = (int32_t)i;
                p.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                p.led_tag[slot] = (uint32_t)agent_tick;
                p.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(p.led_obs + slot * kObsWords);
#pragma unroll
                for (int g = 0; g < kObsWords / 4; ++g)
                    lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            d = r;
#pragma unroll
            for (int j = 0; j < kObsWords; ++j) o[j] = ro[j];
        }
        if (lane == 0) p.led_count[tile] = led_fill + (uint32_t)__popcll(m);
    }

    store_drone(d, p.state, tile, lane);
    if (lane == 0) p.tick[tile] = agent_tick + 1ull;
    if (live) {
        p.reward[i] = reward;
        p.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        p.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
    if constexpr (FUSED) {
        // VecNormalize bookkeeping in the same pass (train_hover.py:43): done mask, discounted return
        // ret = ret * gamma + r (cleared at episode ends), and this tile's fp64 moment sums of the
        // returned observations and of ret — merged across tiles by moments_merge_tiles.
        T* lds_w = reinterpret_cast<T*>(lds);
        if (live && p.done) p.done[i] = done ? 1 : 0;
        T ret = T(0);
        if (p.ret) {
            ret = live ? ret_in * (T)p.gamma + reward : T(0);
            if (live) p.ret[i] = done ? T(0) : ret;
        }
        lds_w[kTile * kObsWords + lane] = ret;
        store_obs_tile(o, p.obs, tile, lane, p.n, lds);   // writes the tile and barriers before reading it back
        if (p.partials) {
            if constexpr (sizeof(T) == 4) {
                const int rows = (int)((p.n - tile * kTile) < kTile ? (p.n - tile * kTile) : kTile);
                tile_moment_sums<kObsWords>(reinterpret_cast<const float*>(lds), reinterpret_cast<double*>(lds_w + kTile * kObsWords + kTile),
                                 rows, lane, TPB > 1 ? bsum + wv * kPartialWords : p.partials + tile * kPartialWords);
                if constexpr (TPB == 4) {
                    __syncthreads();
                    if (wv == 0 && lane < kPartialWords) {
                        const double v = (bsum[lane] + bsum[kPartialWords + lane]) + (bsum[2 * kPartialWords + lane] + bsum[3 * kPartialWords + lane]);
                        double* row = p.partials + (int64_t)blockIdx.x * kPartialWords + lane;
#ifdef FPV_EXPERIMENTS
                        if (p.mg_ticket) row_store_sc1(row, v);      // write-through: read by the last workgroup of this launch
                        else
#endif
                        *row = v;
                    }
#ifdef FPV_EXPERIMENTS                                                  // (merge_in_step: measured slower)
                    if (p.mg_ticket) {                                  // uniform
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        step_merge_tail(p, p.partials, kObsWords, bsum);
                    }
#endif
                }
            }
        }
    } else {
        store_obs_tile(o, p.obs, tile, lane, p.n, lds);
    }
}

// ---------------------------------------------------------------- K1 specialised: physics wave + helper wave
// For drone counts that leave SIMDs empty (<= 32 768 drones = 512 tiles on 1 024 SIMDs) this variant gives every
// 64-drone tile a second wave, which then runs on a SIMD of its own (once the two waves have to share a SIMD the
// instruction mix already saturates its issue rate and nothing is gained — measured, see g_pair_max_tiles):
//   wave P (lanes 0..63)   : state, control, physics, observation, reward, ledger, reset — everything state-bound;
//   wave H (lanes 64..127) : the motor-noise draws (Philox + Box-Muller are state-independent: H runs one sub-step
//                            pair ahead through a double-buffered LDS slot), then the row-major store of the staged
//                            observation tile and, when FUSED, VecNormalize's tile moment sums.
// Seven workgroup barriers pair the two waves.  Arithmetic is identical to step_kernel (same device functions, same
// noise values), so results are bit-identical to it.
template <bool FUSED>
__global__ __launch_bounds__(2 * kTile) void step_kernel_pair(StepArgs<float> p) {
    using T = float;
    __shared__ T nz[2][8][kTile];
    __shared__ Vec4<T> lds[kTile * kObsWords / 4 + (FUSED ? kTile / 4 : 0)];
    __shared__ __attribute__((aligned(32))) float pix[OneWavePixT<float>::kWords];     // pixel detector: the physics wave's tile area
    const int lane = threadIdx.x & (kTile - 1);
    const bool helper = threadIdx.x >= kTile;
    const int64_t tile = blockIdx.x;
    const int64_t i = tile * kTile + lane;
    const bool live = i < p.n;
    const uint64_t agent_tick = p.tick[tile];
    const uint32_t gid = (uint32_t)(p.id_offset + (uint64_t)i);

    if (helper) {
        // ---- noise producer: pair k is published before barrier k
#pragma unroll 1
        for (int k = 0; k < kSubstepPairs; ++k) {
            T ga[4], gb[4];
            normal8(p.seed, gid, agent_tick * (uint64_t)kSubstepPairs + (uint64_t)k, ga, gb);
            T(*slot)[kTile] = nz[k & 1];
            slot[0][lane] = ga[0]; slot[1][lane] = ga[1]; slot[2][lane] = ga[2]; slot[3][lane] = ga[3];
            slot[4][lane] = gb[0]; slot[5][lane] = gb[1]; slot[6][lane] = gb[2]; slot[7][lane] = gb[3];
            __syncthreads();
        }
        __syncthreads();   // barrier 6: wave P has staged the observation tile (and the returns)
        // ---- row-major store of the tile: five 1-KiB contiguous stores
        Vec4<T>* out = reinterpret_cast<Vec4<T>*>(p.obs + tile * (kTile * kObsWords));
        const int64_t rows_left = p.n - tile * kTile;
#pragma unroll
        for (int r = 0; r < kObsWords / 4; ++r) {
            const int f4 = r * kTile + lane;
            if (f4 / (kObsWords / 4) < rows_left) out[f4] = lds[f4];
        }
        if constexpr (FUSED) {
            if (p.partials) {
                const int rows = (int)(rows_left < kTile ? rows_left : kTile);
                tile_moment_sums<kObsWords>(reinterpret_cast<const float*>(lds), reinterpret_cast<double*>(&nz[0][0][0]), rows, lane,
                                 p.partials + tile * kPartialWords);   // the noise slots are free by now: reuse as exchange
            }
        }
        return;
    }

    // ---------------- wave P
    Drone<T> d;
    load_drone(d, p.state, tile, lane);
    const Vec4<T> act = live ? reinterpret_cast<const Vec4<T>*>(p.actions)[i] : Vec4<T>{T(0), T(0), T(0), T(-1)};
    const uint32_t led_fill = p.led_count[tile];
    const T tmpl_lane = p.tmpl[lane < kStateWords + kObsWords ? lane : 0];
    T ret_in = T(0);
    if constexpr (FUSED) {
        if (p.ret && live) ret_in = p.ret[i];
    }
    const T spx = act.x * T(30), spy = act.y * T(30), spz = act.z * T(-30);
    const T thrust = clip_((act.w + T(1)) * T(0.5), T(0), T(1));
    T pwm[4];
#pragma unroll 1
    for (int k = 0; k < kSubstepPairs; ++k) {
        __syncthreads();   // barrier k: pair k's noise is in nz[k & 1]
        T(*slot)[kTile] = nz[k & 1];
        const T ga[4] = {slot[0][lane], slot[1][lane], slot[2][lane], slot[3][lane]};
        const T gb[4] = {slot[4][lane], slot[5][lane], slot[6][lane], slot[7][lane]};
        control_tick(d, spx, spy, spz, thrust, pwm);
        physics_substep<T, true>(d, pwm, p.sigma, ga);
        physics_substep<T, true>(d, pwm, p.sigma, gb);
    }
    T roll, pitch, yaw;
    euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
    T o[kObsWords];
    const bool vis = observe_tile(d, roll, pitch, yaw, act.x, act.y, act.z, act.w, o, p.detector, pix, lane);
    uint32_t bits;
    const T reward = reward_and_flags(d, roll, pitch, vis, o, act.x, act.y, act.z, act.w, p.floor_check != 0, bits);
    d.er = roll; d.ep = pitch; d.eyaw = yaw;
    d.steps += T(1);
    d.a0 = act.x; d.a1 = act.y; d.a2 = act.z; d.a3 = act.w;
    d.ep_ret += reward;

    const bool done = live && (bits & (kInfoTerm | kInfoTrunc)) != 0;
    const unsigned long long m = __ballot(done);
    if (m != 0ull) {
        Drone<T> r = d;
        drone_from_template(r, tmpl_lane);
        T ro[kObsWords];
#pragma unroll
        for (int j = 0; j < kObsWords; ++j) ro[j] = __shfl(tmpl_lane, kStateWords + j);
        if (done) {
            const int64_t row = (int64_t)led_fill + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (row < p.led_rows) {
                const int64_t slot = tile * p.led_rows + row;
                p.led_idx[slot] = (int32_t)i;
                p.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                p.led_tag[slot] = (uint32_t)agent_tick;
                p.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(p.led_obs + slot * kObsWords);
#pragma unroll
                for (int g = 0; g < kObsWords / 4; ++g) lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            d = r;
#pragma unroll
            for (int j = 0; j < kObsWords; ++j) o[j] = ro[j];
        }
        if (lane == 0) p.led_count[tile] = led_fill + (uint32_t)__popcll(m);
    }
    // stage the observation tile (and the discounted returns) for wave H, then finish the state-bound stores
#pragma unroll
    for (int g = 0; g < kObsWords / 4; ++g)
        lds[lane * (kObsWords / 4) + g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
    if constexpr (FUSED) {
        T ret = T(0);
        if (p.ret) {
            ret = live ? ret_in * (T)p.gamma + reward : T(0);
            if (live) p.ret[i] = done ? T(0) : ret;
        }
        reinterpret_cast<T*>(lds)[kTile * kObsWords + lane] = ret;
        if (live && p.done) p.done[i] = done ? 1 : 0;
    }
    __syncthreads();   // barrier 6
    store_drone(d, p.state, tile, lane);
    if (lane == 0) p.tick[tile] = agent_tick + 1ull;
    if (live) {
        p.reward[i] = reward;
        p.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        p.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
}

// ---------------------------------------------------------------- K2 masked reset
template <typename T>
__global__ __launch_bounds__(kTile) void reset_kernel(Vec4<T>* state, const T* tmpl, const uint8_t* mask, T* obs,
                                                      int64_t n) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t i = tile * kTile + lane;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    Drone<T> d;
    load_drone(d, state, tile, lane);
    const T* tm = tmpl;
    d.px = tm[0]; d.py = tm[1]; d.pz = tm[2]; d.steps = tm[3];
    d.qx = tm[4]; d.qy = tm[5]; d.qz = tm[6]; d.qw = tm[7];
    d.vx = tm[8]; d.vy = tm[9]; d.vz = tm[10]; d.ep_ret = tm[11];
    d.wx = tm[12]; d.wy = tm[13]; d.wz = tm[14]; d.prev_area = tm[15];
    d.t0 = tm[16]; d.t1 = tm[17]; d.t2 = tm[18]; d.t3 = tm[19];
    d.ix = tm[20]; d.iy = tm[21]; d.iz = tm[22]; d.prev_ratio = tm[23];
    d.ex = tm[24]; d.ey = tm[25]; d.ez = tm[26]; d.prev_cx = tm[27];
    d.er = tm[28]; d.ep = tm[29]; d.eyaw = tm[30]; d.prev_cy = tm[31];
    store_drone(d, state, tile, lane);
    if (obs) {
        T* o = obs + i * kObsWords;
#pragma unroll
        for (int j = 0; j < kObsWords; ++j) o[j] = tmpl[kStateWords + j];
    }
}

// reset template = spawn state + noise-free motor warm-up + first observation (SPEC §8);
// evaluated once at fpv_create with the same device functions the step uses.
template <typename T>
__global__ __launch_bounds__(kTile) void template_kernel(T* tmpl, int detector) {
    __shared__ __attribute__((aligned(32))) float pix[OneWavePixT<T>::kWords];
    const int lane = threadIdx.x;                          // one wave; every lane evaluates the same spawn state, lane 0 writes
    Drone<T> d = {};
    d.pz = C<T>::z_rest;
    d.qw = T(1);
    T thr = T(0);
    for (int k = 0; k < 20; ++k) thr += C<T>::alpha * (C<T>::pwm_idle - thr);  // hover.py:109-110
    d.t0 = d.t1 = d.t2 = d.t3 = thr;
    T o[kObsWords];
    observe_tile(d, T(0), T(0), T(0), T(0), T(0), T(0), T(0), o, detector, pix, lane);     // hover.py:112-113
    if (lane != 0) return;
    const T s[kStateWords] = {d.px, d.py, d.pz, d.steps, d.qx, d.qy, d.qz, d.qw, d.vx, d.vy, d.vz, d.ep_ret,
                              d.wx, d.wy, d.wz, d.prev_area, d.t0, d.t1, d.t2, d.t3, d.ix, d.iy, d.iz,
                              d.prev_ratio, d.ex, d.ey, d.ez, d.prev_cx, d.er, d.ep, d.eyaw, d.prev_cy,
                              d.a0, d.a1, d.a2, d.a3};
    for (int j = 0; j < kStateWords; ++j) tmpl[j] = s[j];
    for (int j = 0; j < kObsWords; ++j) tmpl[kStateWords + j] = o[j];
}

// fill every storage tile (padding lanes included) with the template
template <typename T>
__global__ void fill_kernel(Vec4<T>* state, const T* tmpl, int64_t tiles) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // granule index
    if (g >= tiles * kGranules * kTile) return;
    const int gran = (int)((g / kTile) % kGranules);
    state[g] = Vec4<T>{tmpl[4 * gran], tmpl[4 * gran + 1], tmpl[4 * gran + 2], tmpl[4 * gran + 3]};
}

// canonical [n][36] <-> tiled granule storage
template <typename T, bool TO_CANON>
__global__ void convert_kernel(Vec4<T>* state, T* canon, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * kStateWords) return;
    const int64_t i = e / kStateWords;
    const int slot = (int)(e % kStateWords);
    T* word = reinterpret_cast<T*>(state + (i / kTile) * (kGranules * kTile) + (slot / 4) * kTile + (i % kTile)) +
              (slot % 4);
    T* c = canon + i * kStateWords + canon_of_slot(slot);
    if (TO_CANON) *c = *word;
    else *word = *c;
}

__global__ void tick_fill_kernel(uint64_t* tick, int64_t tiles, uint64_t v) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < tiles) tick[t] = v;
}

// ---------------------------------------------------------------- launchers
int64_t g_pair_max_tiles = 512;   // step_kernel_pair up to 32 768 drones: P and H waves then land on different SIMDs
                                  // (measured: -14 % at 4 096, -5 % at 32 768, tie at 65 536, +30 % at 131 072)

// step_quad_kernel (fpv_rollstep.hip): four tiles per 512-thread workgroup, a physics and a helper wave per tile that share the
// pixel detector's rows (fp32, motor noise, tile count a multiple of 4).  It pays with the pixel detector while the grid is at
// most one workgroup per CU — us per launch, default dispatch -> this kernel (tools/step_quad_sweep.py,
// profiles/r03_step_quad_sweep.json): 16 384 drones 16.0 -> 15.5, 32 768 17.5 -> 16.0, 65 536 19.3 -> 17.2 (fused 21.1 -> 17.8);
// 131 072 26.7 -> 31.2 (100 KB of LDS per workgroup: one workgroup per CU); with the analytic detector it loses at every size
// (65 536: 12.2 -> 13.1).  Same bits either way.
// Round 4, with the contour pipeline restated in full: the window is what
// profiles/r04_step_quad_sweep.json measured — 1 024 drones 28.7 -> 22.9 us, 4 096 32.2 -> 25.8, 65 536 38.3 -> 29.6; 131 072 50.4 -> 52.8,
// 524 288 141 -> 186 (beyond one workgroup per CU the plain kernel's eight waves per CU hide the contour chain better than four
// tiles' pooled lane groups behind 100 KB of LDS) — hence every multiple of 256 up to 65 536.
int64_t g_quad_min_tiles = 0, g_quad_max_tiles = 1024;        // (0, 65 536] drones
static bool use_quad(int64_t tiles, bool noise, bool f32, int detector) {
    return f32 && noise && detector != 0 && tiles % 4 == 0 && tiles > g_quad_min_tiles && tiles <= g_quad_max_tiles;
}

template <typename T>
hipError_t launch_step(const StepArgs<T>& a_in, bool noise, hipStream_t s, bool* merged) {
    StepArgs<T> a = a_in;
    const int64_t tiles = (a.n + kTile - 1) / kTile;
    if constexpr (sizeof(T) == 4) {
        if (use_quad(tiles, noise, true, a.detector)) {
            a.mg_ticket = nullptr;
            if (merged) *merged = false;
            return launch_step_quad(a, nullptr, s);
        }
    }
    const bool fused = a.done || a.ret || a.partials;
    // the in-launch statistics merge exists in the four-tiles-per-workgroup variant; the reducer holds <= 1 024 rows' worth
    // of sums per thread-way comfortably (tiles / 4 <= 4 096)
    const bool pair = sizeof(T) == 4 && noise && g_pair_max_tiles > 0 && tiles <= g_pair_max_tiles;
    const bool can_merge = sizeof(T) == 4 && fused && a.partials && a.mg_ticket && !pair && tiles % 4 == 0 && tiles / 4 <= 4096;
#ifndef FPV_EXPERIMENTS
    a.mg_ticket = nullptr;                        // the in-launch merge is an FPV_EXPERIMENTS build's
    if (merged) *merged = false;
#else
    if (!can_merge) a.mg_ticket = nullptr;
    if (merged) *merged = can_merge;
#endif
    if constexpr (sizeof(T) == 4) {
        // few enough tiles that the chip would sit at one wave per SIMD: physics wave + helper wave per tile
        if (noise && g_pair_max_tiles > 0 && tiles <= g_pair_max_tiles) {
            if (fused) hipLaunchKernelGGL((step_kernel_pair<true>), dim3((unsigned)tiles), dim3(2 * kTile), 0, s, a);
            else hipLaunchKernelGGL((step_kernel_pair<false>), dim3((unsigned)tiles), dim3(2 * kTile), 0, s, a);
            return hipGetLastError();
        }
        if (fused) {
            if (tiles % 4 == 0) {               // four tiles per workgroup, one partial row per workgroup (step_partial_rows)
                if (noise) hipLaunchKernelGGL((step_kernel<T, true, true, 4>), dim3((unsigned)(tiles / 4)), dim3(4 * kTile), 0, s, a);
                else hipLaunchKernelGGL((step_kernel<T, false, true, 4>), dim3((unsigned)(tiles / 4)), dim3(4 * kTile), 0, s, a);
                return hipGetLastError();
            }
            if (noise) hipLaunchKernelGGL((step_kernel<T, true, true>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
            else hipLaunchKernelGGL((step_kernel<T, false, true>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
            return hipGetLastError();
        }
    }
    if (noise) hipLaunchKernelGGL((step_kernel<T, true, false>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
    else hipLaunchKernelGGL((step_kernel<T, false, false>), dim3((unsigned)tiles), dim3(kTile), 0, s, a);
    return hipGetLastError();
}
// rows of extras.partials that a fused fp32 step of n drones writes (mirrors the dispatch above)
int64_t step_partial_rows(int64_t n, bool noise, bool pixel) {
    const int64_t tiles = (n + kTile - 1) / kTile;
    if (use_quad(tiles, noise, true, pixel ? 1 : 0)) return tiles / 4;
    if (noise && g_pair_max_tiles > 0 && tiles <= g_pair_max_tiles) return tiles;
    return tiles % 4 == 0 ? tiles / 4 : tiles;
}

template hipError_t launch_step<float>(const StepArgs<float>&, bool, hipStream_t, bool*);
template hipError_t launch_step<double>(const StepArgs<double>&, bool, hipStream_t, bool*);

template <typename T>
hipError_t launch_reset(void* state, const void* tmpl, const uint8_t* mask, void* obs, int64_t n, hipStream_t s) {
    const int64_t tiles = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(reset_kernel<T>, dim3((unsigned)tiles), dim3(kTile), 0, s, (Vec4<T>*)state, (const T*)tmpl,
                       mask, (T*)obs, n);
    return hipGetLastError();
}
template hipError_t launch_reset<float>(void*, const void*, const uint8_t*, void*, int64_t, hipStream_t);
template hipError_t launch_reset<double>(void*, const void*, const uint8_t*, void*, int64_t, hipStream_t);

template <typename T>
hipError_t launch_init(void* state, void* tmpl, int64_t n, int detector, hipStream_t s) {
    const int64_t tiles = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(template_kernel<T>, dim3(1), dim3(64), 0, s, (T*)tmpl, detector);
    const int64_t g = tiles * kGranules * kTile;
    hipLaunchKernelGGL(fill_kernel<T>, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, s, (Vec4<T>*)state,
                       (const T*)tmpl, tiles);
    return hipGetLastError();
}
template hipError_t launch_init<float>(void*, void*, int64_t, int, hipStream_t);
template hipError_t launch_init<double>(void*, void*, int64_t, int, hipStream_t);

template <typename T>
hipError_t launch_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s) {
    const int64_t e = n * kStateWords;
    const dim3 grid((unsigned)((e + 255) / 256)), block(256);
    if (to_canon) hipLaunchKernelGGL((convert_kernel<T, true>), grid, block, 0, s, (Vec4<T>*)state, (T*)canon, n);
    else hipLaunchKernelGGL((convert_kernel<T, false>), grid, block, 0, s, (Vec4<T>*)state, (T*)canon, n);
    return hipGetLastError();
}
template hipError_t launch_convert<float>(void*, void*, int64_t, bool, hipStream_t);
template hipError_t launch_convert<double>(void*, void*, int64_t, bool, hipStream_t);

hipError_t launch_tick_fill(uint64_t* tick, int64_t tiles, uint64_t v, hipStream_t s) {
    hipLaunchKernelGGL(tick_fill_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, s, tick, tiles, v);
    return hipGetLastError();
}

}  // namespace fpv
