// fpv_rollstep.hip — one launch per rollout step: actor-critic forward AND the env step of the same 256 drones (gfx950).
//
// The PPO rollout's step is policy(obs_t) -> a_t -> env.step(a_t) -> obs_{t+1} (SB3 collect_rollouts over
// VecNormalize(VecEnv), simulation/train_hover.py:42-60).  As two kernels (fpv_policy.hip, fpv_step.hip) the action
// and the observation take a trip through HBM between them, the env-step launch runs one 64-drone wave per SIMD
// (issue-bound, profiles/r02_step_instruction_mix.json) and every step pays two launches.  Here the 512-thread
// workgroup that has just evaluated pi and V for its 256 drones steps those same drones:
//   phase 1 (all 8 waves, 32 drones each): fpv_policy.hip's kernel body — VecNormalize merge prologue, normalise,
//            pi / V on v_mfma_f32_32x32x2_f32, sample, log-prob, clip.  Same arithmetic, same bits.
//   phase 2 (the weight image in LDS is dead by now; its space carries the exchange):
//            waves 0..3 = one 64-drone storage tile each: control, physics, observation, reward, ledger, reset;
//            waves 4..7 = their helpers: motor-noise draws one sub-step pair ahead, the row-major observation
//            store and VecNormalize's moment sums — fpv_step.hip's wave-pair split, here with both waves of a pair on
//            the same SIMD, where two waves issue twice what a lone one does (profiles/r02_valu_rate.txt).
// The env-step arithmetic is fpv_step.hip's (same device functions, -ffp-contract=on), so states, observations,
// rewards and ledgers are bit-identical to the two-kernel path; the workgroup writes ONE row of moment sums like
// step_kernel<.., 4>.  The statistics rows of step t are read by every workgroup of launch t + 1 while launch t + 1
// writes its own rows: the caller alternates between two row tables.
//
// Reference path replaced: QuadXHoverEnv.step()/reset() simulation/hover.py:72-114,334-358 (rollout_step_kernel<0>) and the yaw
// task's step simulation/yaw.py:76-149 (rollout_step_kernel<1>) under SB3's VecEnv auto-reset, together with MlpPolicy.forward +
// VecNormalize of simulation/train_hover.py:43-58.
#include "fpv_mlp.h"
#include "fpv_policy_image.h"
#pragma clang fp contract(fast)
#ifndef FPV_ROLLSTEP_KEEP_SPLIT_FENCE
#define FPV_SPLIT_FENCE()            // the rollout-step kernel schedules its k-steps freely (fpv_policy_split.h; A/B: -DFPV_ROLLSTEP_KEEP_SPLIT_FENCE)
#endif
// Round 5, second session (same-box A/B, profiles/r05_split_kernel_ab.txt: 57.22 / 57.25 -> 56.69 / 56.67 us per step with both, the
// same bits): V's image by LDS-DMA in flight under pi's head arithmetic, and layer 2 as a hand-placed stream in THIS translation
// unit's kernels (fpv_policy_split.h; the stand-alone kernel keeps the compiler's order).  -DFPV_RSS_DMA_V=0 / -DFPV_SPLIT_HAND=0: A/B.
#ifndef FPV_RSS_DMA_V
#define FPV_RSS_DMA_V 1
#endif
#ifndef FPV_SPLIT_HAND
#define FPV_SPLIT_HAND 1
#endif
#ifndef FPV_RSS_DMA_PI
#define FPV_RSS_DMA_PI 1              // pi's image by LDS-DMA ahead of the statistics merge (57.73 / 57.69 -> 57.53 / 57.45 us on another box; =0: A/B)
#endif
#include "fpv_policy_split.h"   // (the split policy arithmetic is compiled as in fpv_policy_split.hip: contract fast)
#pragma clang fp contract(on)
#include "fpv_pixel_tile.h"
#include "fpv_step_parts.h"
#include "fpv_yaw_parts.h"

namespace fpv {

struct RollStepArgs {
    // ---- policy side (fpv_policy_io)
    const float* image;      // packed weights (kImageFloats; the split image for the split kernels), the task's widths (20 / 4 or 12 / 1)
    const double* rms;       // VecNormalize statistics BEFORE this launch's merge, or null
    float* obs_n;            // [n][20] normalised observation out (rollout buffer), may be null
    float* action;           // [n][4] sampled action, may be null
    float* clipped;          // [n][4] clip(action, -1, 1), may be null (the env step takes it from LDS)
    float* value;            // [n]
    float* logp;             // [n], may be null
    uint64_t pol_seed, pol_id_offset;
    float clip_obs, eps_norm;
    MergeArgs mg;            // mg.rows != null: fold the previous launch's rows into rms first
    // ---- env side: st.obs holds obs_t on entry and obs_{t+1} on exit; st.actions is not used
    StepArgs<float> st;
    uint64_t* stamps;        // diagnostic (fpv_debug_rollout_stamps): [grid][16] s_memrealtime values, or null
};

// LDS map of phase 2, in floats from the start of the (dead) weight image
constexpr int kRsAct = 0;                                   // f32x4 [256] clipped actions
constexpr int kRsNoise = 1024;                              // float [4 tiles][2 slots][8][64]
constexpr int kRsTile = kRsNoise + 4 * 2 * 8 * kTile;       // per tile: [64][20] observations + 64 returns
constexpr int kRsTileFloats = kTile * kObsWords + kTile;
constexpr int kRsXch = kRsTile + 4 * kRsTileFloats;         // per tile: 128 doubles of moment exchange
constexpr int kRsSum = kRsXch + 4 * 256;                    // 4 x kPartialWords doubles
static_assert((kRsXch * 4) % 8 == 0 && (kRsSum * 4) % 8 == 0, "fp64 areas must be 8-byte aligned");
static_assert(kRsSum + 2 * 4 * kPartialWords <= 2 * kNetFloats, "phase-2 exchange must fit in the weight image's space");
// Pixel detector (hover.py:157-222, SPEC §6b): the blobs of the workgroup's four tiles are worked by lane groups of all eight waves
// (fpv_pixel_tile.h).  Above the rest of the phase-2 map: per tile corner block and results, then the schedule and per octet of
// lanes a span table and a contour list.
using RsPix = PixPool<float, 4, 8>;
constexpr int kRsPix = kRsSum + 2 * 4 * kPartialWords;
static_assert((kRsPix * 4) % 16 == 0, "16-byte areas must be 16-byte aligned");
static_assert(kRsPix + RsPix::kWords <= 2 * kNetFloats, "the pixel detector's exchange must fit in the weight image's space");

// Workgroup barrier that publishes LDS only: waits for this wave's LDS operations, not for its global stores.  __syncthreads()
// drains vmcnt too, and in these kernels every barrier would then wait out the ~1-2 us of the rollout-buffer / ledger / flag
// stores issued just before it.  Nothing here passes GLOBAL data between waves of a workgroup across a barrier (where a row is
// handed to other workgroups the storing wave drains it explicitly).
// words per workgroup of the diagnostic stamp buffer (fpv_debug_rollout_stamps); the contour probe build (tools/contour_probe.py,
// -DFPV_CONTOUR_PROBE=1: every wave's contour times, and the contour phase run a second time on warm caches) takes 64
#ifdef FPV_CONTOUR_PROBE
constexpr int kStampStride = 64;
#else
constexpr int kStampStride = 16;
#endif
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- phase 1: fpv_policy.hip's body for one 256-drone chunk
#pragma clang fp contract(fast)
// per-step destinations of the policy outputs (each may be null except value)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kHalf = kIn / 2;                              // 10: words of a row held by one lane half (k = 10 h + j)

struct PolicyOut {
    float* obs_n; float* action; float* clipped; float* value; float* logp;
};

// weight-image floats [first, first + count) -> the same offsets of LDS; count a multiple of 4.  Every granule of a
// thread is requested before the first one is written.
template <int kMaxPer>
__device__ __forceinline__ void rs_stage(const float* __restrict__ image, float* lds, int first, int count) {
    const f32x4* src = reinterpret_cast<const f32x4*>(image + first);
    f32x4* dst = reinterpret_cast<f32x4*>(lds + first);
    const int gran = count / 4;
    f32x4 stage[kMaxPer];
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
        const int g = q * 512 + (int)threadIdx.x;
        stage[q] = src[g < gran ? g : 0];
    }
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
        const int g = q * 512 + (int)threadIdx.x;
        if (g < gran) dst[g] = stage[q];
    }
}

// normalise -> pi, V -> sample -> log-prob -> clip for this lane's drone d (both halves of the wave share it); xraw = the raw
// observation words (IN / 2) h .. (IN / 2) h + IN / 2 - 1 of d.  IN / ACT: 20 / 4 hover, 12 / 1 yaw.  Weights and normaliser
// are in LDS.
// lane = threadIdx.x & 63, passed in so that a caller with a step loop can hide it from loop-invariant code motion (hipcc
// otherwise precomputes every LDS address of the unrolled layers outside the loop, one register each).
template <int IN, int ACT>
__device__ __forceinline__ f32x4 rs_policy_core(const float* lds, const f32x2 (&xraw)[IN / 4], int64_t d, bool live, bool normalise,
                                                float clip_obs, const PolicyOut& out, uint64_t tick, uint64_t pol_seed,
                                                uint64_t pol_id_offset, int lane) {
    constexpr int HALF = IN / 2;
    const float* hb = lds + 2 * kNetFloats;
    const int h = lane >> 5;
    const float clip = normalise ? clip_obs : 3.0e38f;
    float x[HALF];
    {
#pragma unroll
        for (int q = 0; q < HALF / 2; ++q) {
            f32x2 v = xraw[q];
            if (!live) v = f32x2{0.f, 0.f};
            x[2 * q] = v.x; x[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int k = HALF * h + j;
            x[j] = fminf(fmaxf((x[j] - hb[660 + k]) * hb[680 + k], -clip), clip);
        }
        if (out.obs_n && live) {
            f32x2* o = reinterpret_cast<f32x2*>(out.obs_n + d * IN + HALF * h);
#pragma unroll
            for (int q = 0; q < HALF / 2; ++q) o[q] = f32x2{x[2 * q], x[2 * q + 1]};
        }
    }
    float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int net = 0; net < 2; ++net) {
        const float* nb = lds + net * kNetFloats;
        Hidden a;
        bias_init(a, nb + kOffB1, h);
#pragma unroll
        for (int s = 0; s < HALF; ++s)             // IN / 2 products per tile: no padded k (the image keeps 12 slots per tile)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[kOffW1 + (m * 12 + s) * 64 + lane], x[s], a.t[m], 0, 0, 0);
        carrier_all_prescaled(a);
        Hidden c;
        bias_init(c, nb + kOffB2, h);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 w[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    w[m] = *reinterpret_cast<const f32x4*>(nb + kOffW2 + ((((m * 4 + kt) * 4 + r4) * 64 + lane) * 4));
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        c.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[m][rr], a.t[kt][4 * r4 + rr], c.t[m], 0, 0, 0);
            }
        carrier_all_prescaled(c);
        if (net == 0) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(hb + (((h * 4 + kt) * 16 + r) * 4));
                    const float v = c.t[kt][r];
                    head[0] += w.x * v; head[1] += w.y * v; head[2] += w.z * v; head[3] += w.w * v;
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) head[4] += hb[512 + (h * 4 + kt) * 16 + r] * c.t[kt][r];
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) head[j] = half_sum(head[j]);
    if (live && h == 1) out.value[d] = head[4] + hb[644];
    const float mu[4] = {head[0] + hb[640], head[1] + hb[641], head[2] + hb[642], head[3] + hb[643]};

    // sample: Philox4x32-10 keyed like fpv_policy.hip (stream tag 'POL1')
    float eps[4];
    {
        uint32_t xr[4];
        philox4x32_10((uint32_t)(pol_id_offset + (uint64_t)d), (uint32_t)tick, (uint32_t)(tick >> 32), 0x504F4C31u,
                      (uint32_t)pol_seed, (uint32_t)(pol_seed >> 32), xr);
        constexpr float k24 = 5.9604644775390625e-8f, m2ln2 = -1.3862943611198906f;
        const float u0 = ((float)(xr[0] >> 8) + 0.5f) * k24, u1 = ((float)(xr[1] >> 8) + 0.5f) * k24;
        const float u2 = ((float)(xr[2] >> 8) + 0.5f) * k24, u3 = ((float)(xr[3] >> 8) + 0.5f) * k24;
        const float ra = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u0)), rb = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u2));
        eps[0] = ra * __builtin_amdgcn_cosf(u1); eps[1] = ra * __builtin_amdgcn_sinf(u1);
        eps[2] = rb * __builtin_amdgcn_cosf(u3); eps[3] = rb * __builtin_amdgcn_sinf(u3);
    }
    float act[4], lp = hb[656];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        act[j] = mu[j] + hb[652 + j] * eps[j];
        if (j < ACT) lp -= 0.5f * eps[j] * eps[j];
    }
    const f32x4 cl = f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f),
                           fminf(fmaxf(act[2], -1.f), 1.f), fminf(fmaxf(act[3], -1.f), 1.f)};
    if (live) {
        if (h == 0) {
            if constexpr (ACT == 4) {
                if (out.action) reinterpret_cast<f32x4*>(out.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
                if (out.clipped) reinterpret_cast<f32x4*>(out.clipped)[d] = cl;
            } else {
#pragma unroll
                for (int j = 0; j < ACT; ++j) {
                    if (out.action) out.action[d * ACT + j] = act[j];
                    if (out.clipped) out.clipped[d * ACT + j] = cl[j];
                }
            }
        } else if (out.logp) {
            out.logp[d] = lp;
        }
    }
    return cl;
}

// staging, merge prologue and the observation rows (from memory) of the one-step kernel; ends with a workgroup barrier
template <int IN>
__device__ __forceinline__ void rollstep_prologue(const RollStepArgs& p, float* lds, f32x2 (&xpre)[IN / 4]) {
    float* hb = lds + 2 * kNetFloats;                // heads [660], then normaliser mean[20], inv[20]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int64_t d = (int64_t)blockIdx.x * 256 + wave * 32 + col;
    {
        const f32x2* row = reinterpret_cast<const f32x2*>(p.st.obs + (d < p.st.n ? d : 0) * IN + (IN / 2) * h);
#pragma unroll
        for (int q = 0; q < IN / 4; ++q) xpre[q] = row[q];
    }
    MergeRegs mr;
    if (p.mg.rows) merge_request(p.mg, mr);
    rs_stage<(kImageFloats / 4 + 511) / 512>(p.image, lds, 0, kImageFloats);
    if (!p.mg.rows && (int)threadIdx.x < IN) {
        hb[660 + threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
        hb[680 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[IN + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
    }
    if (p.mg.rows) merge_prologue(p.mg, mr, p.rms, reinterpret_cast<double*>(hb + 700), p.eps_norm, hb + 660, hb + 680);
    __syncthreads();
}
#pragma clang fp contract(on)

// ---------------------------------------------------------------- phase 2: the env step of the workgroup's four tiles
// Per-step destinations and switches of phase 2.
struct TailOut {
    float* reward; uint8_t* terminated; uint8_t* truncated; uint8_t* done;   // [n] each (done may be null)
    double* partials;        // this step's row table (one row per workgroup) or null
    bool store_obs;          // write the row-major observation tile to st.obs
    bool row_write_through;  // publish the row with sc1 stores + drain (read by other workgroups inside this launch)
    bool store_state;        // the physics wave stores its tile's state and tick itself, ahead of barrier 6 (one-step kernel)
    uint64_t* stamps;        // diagnostic: s_memrealtime values of tile 0's waves, or null: [0..3] physics wave: pairs done, observed,
                             // barrier 6, 7; pixel detector: [4] corners projected, border pre-check done (before barrier P), [6] P
                             // passed, [7] its groups' blobs done, [8] barrier R passed; helper wave: [10] attitude converted
                             // (before barrier P), [11] its groups' blobs done
};

// Physics-wave registers that live across steps in the persistent kernel; the one-step kernel loads / stores them around
// one call.
struct TileRegs {
    Drone<float> d;
    uint64_t agent_tick;
    uint32_t led_fill;
    float tmpl_lane;
};

// Called by all 512 threads after the clipped actions are in LDS (kRsAct) and a workgroup barrier.  Contains eight (twelve with the
// pixel detector) workgroup barriers on every path.  On return the four observation tiles (+ returns) are still in LDS (kRsTile).
// (Measured and not kept: a two-party LDS arrival counter per tile in place of the workgroup barriers between a tile's two waves —
// no waiting for the slowest of the four tiles at every hand-off, but each hand-off then costs an LDS poll: 68.45 vs 68.63 us per
// step, the physics loop 5.9 -> 6.3 us.)
__device__ __forceinline__ void rs_tail(const StepArgs<float>& s, float* lds, TileRegs& tr, const TailOut& o_) {
    using T = float;
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    const int tl = wave & 3;                           // tile of this workgroup served by this wave
    const bool helper = wave >= 4;
    const int64_t tile = (int64_t)blockIdx.x * 4 + tl;
    const int64_t i = tile * kTile + lane;
    const bool live = i < s.n;
    const uint32_t gid = (uint32_t)(s.id_offset + (uint64_t)i);
    T(*nz)[8][kTile] = reinterpret_cast<T(*)[8][kTile]>(lds + kRsNoise + tl * (2 * 8 * kTile));
    Vec4<T>* tl_lds = reinterpret_cast<Vec4<T>*>(lds + kRsTile + tl * kRsTileFloats);
    double* bsum = reinterpret_cast<double*>(lds + kRsSum);
    float* pix = lds + kRsPix;                         // pixel detector: the workgroup's pool (fpv_pixel_tile.h)

    if (helper) {
        // ---- noise producer: pair k is published before barrier k (fpv_step.hip, step_kernel_pair)
#pragma unroll 1
        for (int k = 0; k < kSubstepPairs; ++k) {
            T ga[4], gb[4];
            normal8(s.seed, gid, tr.agent_tick * (uint64_t)kSubstepPairs + (uint64_t)k, ga, gb);
            T(*slot)[kTile] = nz[k & 1];
            slot[0][lane] = ga[0]; slot[1][lane] = ga[1]; slot[2][lane] = ga[2]; slot[3][lane] = ga[3];
            slot[4][lane] = gb[0]; slot[5][lane] = gb[1]; slot[6][lane] = gb[2]; slot[7][lane] = gb[3];
            lds_barrier();
        }
        if (s.detector) {                                  // uniform: the pixel detector's blobs are spread over both waves
            lds_barrier();                                 // barrier P0: attitude and previous Euler angles of the tile's drones are in LDS
            {   // This wave converts the attitude (Euler angles, obs[0:7]) while the physics wave projects the corners.
                f32x4* att = reinterpret_cast<f32x4*>(nz);  // the noise slots are free once pair 5 has been read
                const f32x4 q = att[lane], pe = att[kTile + lane];
                T roll, pitch, yaw, o7[7];
                euler_from_quat(q.x, q.y, q.z, q.w, roll, pitch, yaw);
                observe_attitude(pe.x, pe.y, pe.z, roll, pitch, yaw, o7);
                att[lane] = f32x4{roll, pitch, yaw, o7[0]};
                att[kTile + lane] = f32x4{o7[1], o7[2], o7[3], o7[4]};
                att[2 * kTile + lane] = f32x4{o7[5], o7[6], 0.f, 0.f};
            }
            if (o_.stamps && threadIdx.x == 256) o_.stamps[10] = __builtin_amdgcn_s_memrealtime();
            lds_barrier();                                 // barrier P1 (the owner waves write the schedule behind it)
        }
    }
    // ---------------- physics wave, first part: the sub-steps, then the corners of its drones' panels
    Drone<T>& d = tr.d;
    Vec4<T> act = Vec4<T>{T(0), T(0), T(0), T(-1)};
    T ret_in = T(0);
    if (!helper) {
        const f32x4 av = reinterpret_cast<const f32x4*>(lds + kRsAct)[tl * kTile + lane];
        if (live) act = Vec4<T>{av.x, av.y, av.z, av.w};
        if (s.ret && live) ret_in = s.ret[i];
        const T spx = act.x * T(30), spy = act.y * T(30), spz = act.z * T(-30);
        const T thrust = clip_((act.w + T(1)) * T(0.5), T(0), T(1));
        T pwm[4];
#pragma unroll 1
        for (int k = 0; k < kSubstepPairs; ++k) {
            lds_barrier();   // barrier k: pair k's noise is in nz[k & 1]
            T(*slot)[kTile] = nz[k & 1];
            const T ga[4] = {slot[0][lane], slot[1][lane], slot[2][lane], slot[3][lane]};
            const T gb[4] = {slot[4][lane], slot[5][lane], slot[6][lane], slot[7][lane]};
            control_tick(d, spx, spy, spz, thrust, pwm);
            physics_substep<T, true>(d, pwm, s.sigma, ga);
            physics_substep<T, true>(d, pwm, s.sigma, gb);
        }
        if (o_.stamps && threadIdx.x == 0) o_.stamps[0] = __builtin_amdgcn_s_memrealtime();
        if (s.detector) {
            // Pixel detector (hover.py:157-222, SPEC §6b).  A drone's blob is 0 .. ~60 image rows and a fifth of the drones have one
            // at all: one lane working its own drone's blob keeps the wave busy for the slowest one while most lanes idle, and the
            // helper wave has nothing to do until barrier 6.  So the blobs of the workgroup's four tiles are spread over all EIGHT
            // waves: this wave projects the corners and drops the blobs that cannot be seen whole (exact pre-check) — the helper
            // converts the attitude (Euler angles, obs[0:7]) meanwhile; then every wave's lane groups (8 lanes, 16 for the tall
            // blobs) take a blob each: its rows' spans, cv2's contour, cv2.arcLength, cv2.approxPolyDP (fpv_pixel_tile.h,
            // fpv_polydp.h); this wave reads its drones' results.
            f32x4* att = reinterpret_cast<f32x4*>(nz);        // the noise slots are free once pair 5 has been read
            att[lane] = f32x4{d.qx, d.qy, d.qz, d.qw};
            att[kTile + lane] = f32x4{d.er, d.ep, d.eyaw, 0.f};
            lds_barrier();                                     // barrier P0: the helper converts the attitude from here on
            const int blob_rows = pixel_tile_setup<float, 4, 8>(d, pix, tl, lane);
            lds_barrier();                                     // barrier P1: the four tiles' census words are in LDS
            pixel_tile_schedule<float, 4, 8>(pix, tl, lane, blob_rows);      // this tile's octets of the workgroup's schedule
            if (o_.stamps && threadIdx.x == 0) o_.stamps[4] = __builtin_amdgcn_s_memrealtime();
        }
    }
    // ---------------- every wave: the blobs (one copy of this code for both kinds of wave)
    if (s.detector) {
        lds_barrier();                                         // barrier P: the four tiles' corner blocks are in LDS
        if (o_.stamps && threadIdx.x == 0) o_.stamps[6] = __builtin_amdgcn_s_memrealtime();
#ifdef FPV_CONTOUR_PRIO
        // A SIMD serves its older wave first: with the same instruction stream on both, the helper wave (the younger one) ends a
        // microsecond behind the physics wave, and barrier R waits for it.  FPV_CONTOUR_PRIO raises the helper's priority for
        // the length of the contour phase (A/B switch; DESIGN §5b has the measurement).
        if (helper) __builtin_amdgcn_s_setprio(FPV_CONTOUR_PRIO);
#endif
#ifdef FPV_CONTOUR_PROBE
        if (o_.stamps && lane == 0) o_.stamps[36 + wave] = __builtin_amdgcn_s_memrealtime();                  // s[40 + w]: wave w past P
        pixel_pool_contours<float, 4, 8>(pix, wave, lane, o_.stamps ? o_.stamps + 44 + wave : nullptr);      // s[48 + w]: schedule done
        if (o_.stamps && lane == 0) o_.stamps[12 + wave] = __builtin_amdgcn_s_memrealtime();                  // s[16 + w]: blobs done
        if (o_.stamps) {                                       // the same blobs once more: instructions and tables are warm now
            lds_barrier();
            if (lane == 0) o_.stamps[20 + wave] = __builtin_amdgcn_s_memrealtime();                           // s[24 + w]
            pixel_pool_contours<float, 4, 8>(pix, wave, lane, o_.stamps + 52 + wave);                         // s[56 + w]
            if (lane == 0) o_.stamps[28 + wave] = __builtin_amdgcn_s_memrealtime();                           // s[32 + w]
        }
#else
        pixel_pool_contours<float, 4, 8>(pix, wave, lane);
#endif
#ifdef FPV_CONTOUR_PRIO
        if (helper) __builtin_amdgcn_s_setprio(0);
#endif
        if (o_.stamps && (threadIdx.x & 255) == 0) o_.stamps[helper ? 11 : 7] = __builtin_amdgcn_s_memrealtime();
        lds_barrier();                                         // barrier R: every blob's result is in LDS
        if (o_.stamps && threadIdx.x == 0) o_.stamps[8] = __builtin_amdgcn_s_memrealtime();
    }
    if (helper) {
        lds_barrier();   // barrier 6: the physics wave has staged the observation tile (and the returns)
        const int64_t rows_left = s.n - tile * kTile;
        // the moment sums come first: their in-wave exchange fences would otherwise wait out the tile's global stores
        if (o_.partials) {
            const int rows = (int)(rows_left < kTile ? (rows_left > 0 ? rows_left : 0) : kTile);
            tile_moment_sums<kObsWords>(reinterpret_cast<const float*>(tl_lds), reinterpret_cast<double*>(lds + kRsXch + tl * 256),
                                        rows, lane, bsum + tl * kPartialWords);
        }
        if (o_.store_obs) {
            Vec4<T>* out = reinterpret_cast<Vec4<T>*>(s.obs + tile * (kTile * kObsWords));
#pragma unroll
            for (int r = 0; r < kObsWords / 4; ++r) {
                const int f4 = r * kTile + lane;
                if (f4 / (kObsWords / 4) < rows_left) out[f4] = tl_lds[f4];
            }
        }
        lds_barrier();   // barrier 7: the four tiles' sums are in LDS
        if (o_.partials && wave == 4) {
            if (lane < kPartialWords) {
                const double v = (bsum[lane] + bsum[kPartialWords + lane]) + (bsum[2 * kPartialWords + lane] + bsum[3 * kPartialWords + lane]);
                double* row = o_.partials + (int64_t)blockIdx.x * kPartialWords + lane;
                if (o_.row_write_through) row_store_sc1(row, v);
                else *row = v;
            }
            if (o_.row_write_through) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        tr.agent_tick += 1ull;
        return;
    }

    // ---------------- physics wave, second part
    T roll, pitch, yaw;
    T o[kObsWords];
    bool vis;
    T cam_x, cam_y, cam_area, cam_ratio;
    if (s.detector) {
        f32x4* att = reinterpret_cast<f32x4*>(nz);
        {
            const f32x4 t0 = att[lane], t1 = att[kTile + lane], t2 = att[2 * kTile + lane];
            roll = t0.x; pitch = t0.y; yaw = t0.z;
            o[0] = t0.w; o[1] = t1.x; o[2] = t1.y; o[3] = t1.z; o[4] = t1.w; o[5] = t2.x; o[6] = t2.y;
        }
        vis = pixel_tile_features<float, 4, 8>(pix, tl, lane, cam_x, cam_y, cam_area, cam_ratio);
    } else {
        euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
        T o7[7];
        observe_attitude(d.er, d.ep, d.eyaw, roll, pitch, yaw, o7);
#pragma unroll
        for (int j = 0; j < 7; ++j) o[j] = o7[j];
        vis = camera_features(d, cam_x, cam_y, cam_area, cam_ratio);
    }
    observe_features(d, act.x, act.y, act.z, act.w, o, vis, cam_x, cam_y, cam_area, cam_ratio);
    uint32_t bits;
    const T reward = reward_and_flags(d, roll, pitch, vis, o, act.x, act.y, act.z, act.w, s.floor_check != 0, bits);
    d.er = roll; d.ep = pitch; d.eyaw = yaw;
    d.steps += T(1);
    d.a0 = act.x; d.a1 = act.y; d.a2 = act.z; d.a3 = act.w;
    d.ep_ret += reward;
    if (o_.stamps && threadIdx.x == 0) o_.stamps[1] = __builtin_amdgcn_s_memrealtime();

    const bool done = live && (bits & (kInfoTerm | kInfoTrunc)) != 0;
    const unsigned long long m = __ballot(done);
    if (m != 0ull) {
        Drone<T> r = d;
        drone_from_template(r, tr.tmpl_lane);
        T ro[kObsWords];
#pragma unroll
        for (int j = 0; j < kObsWords; ++j) ro[j] = __shfl(tr.tmpl_lane, kStateWords + j);
        if (done) {
            const int64_t row = (int64_t)tr.led_fill + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (row < s.led_rows) {
                const int64_t slot = tile * s.led_rows + row;
                s.led_idx[slot] = (int32_t)i;
                s.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                s.led_tag[slot] = (uint32_t)tr.agent_tick;
                s.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(s.led_obs + slot * kObsWords);
#pragma unroll
                for (int g = 0; g < kObsWords / 4; ++g) lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            d = r;
#pragma unroll
            for (int j = 0; j < kObsWords; ++j) o[j] = ro[j];
        }
        tr.led_fill += (uint32_t)__popcll(m);
        if (lane == 0) s.led_count[tile] = tr.led_fill;
    }
#pragma unroll
    for (int g = 0; g < kObsWords / 4; ++g)
        tl_lds[lane * (kObsWords / 4) + g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
    {
        T ret = T(0);
        if (s.ret) {
            ret = live ? ret_in * (T)s.gamma + reward : T(0);
            if (live) s.ret[i] = done ? T(0) : ret;
        }
        reinterpret_cast<T*>(tl_lds)[kTile * kObsWords + lane] = ret;
        if (live && o_.done) o_.done[i] = done ? 1 : 0;
    }
    lds_barrier();   // barrier 6 (publishes LDS only: the global stores around it stay in flight)
    if (o_.stamps && threadIdx.x == 0) o_.stamps[2] = __builtin_amdgcn_s_memrealtime();
    // Everything this wave still owes to memory leaves now, under the helper's tile store and moment sums, instead of after
    // barrier 7 at the very end of the kernel: the state's nine 1-KiB stores, the tick, reward and flags.
    tr.agent_tick += 1ull;
    if (o_.store_state) {
        store_drone(d, s.state, tile, lane);
        if (lane == 0) s.tick[tile] = tr.agent_tick;
    }
    if (live) {
        o_.reward[i] = reward;
        o_.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        o_.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
    lds_barrier();   // barrier 7
    if (o_.stamps && threadIdx.x == 0) o_.stamps[3] = __builtin_amdgcn_s_memrealtime();
}

// physics waves: drone state, ledger fill and reset template of the wave's tile; every wave: the tile's agent tick
__device__ __forceinline__ void rs_load_state(const StepArgs<float>& s, TileRegs& tr) {
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (wave & 3);
    tr.led_fill = 0u;
    tr.tmpl_lane = 0.0f;
    if (wave < 4) {
        load_drone(tr.d, s.state, tile, lane);
        tr.led_fill = s.led_count[tile];
        tr.tmpl_lane = s.tmpl[lane < kStateWords + kObsWords ? lane : 0];
    }
}
__device__ __forceinline__ void rs_load_tile(const StepArgs<float>& s, TileRegs& tr) {
    tr.agent_tick = s.tick[(int64_t)blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)];
    rs_load_state(s, tr);
}
__device__ __forceinline__ void rs_store_state(const StepArgs<float>& s, const TileRegs& tr) {
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    if (wave < 4) store_drone(tr.d, s.state, (int64_t)blockIdx.x * 4 + (wave & 3), lane);
}
// ---------------------------------------------------------------- phase 2 of the yaw task (fpv_yaw.hip's step body)
// One aviary step = one control tick + two sub-steps per agent step (yaw.py:126-127): the physics wave draws its own eight
// normals, so there is no noise hand-off; waves 4..7 only take the observation tile (row-major store, moment sums).
// LDS: kRsAct as above, then per tile [64][12] observations + 64 returns, 128 doubles of exchange, 4 x 28 doubles of sums.
constexpr int kRsYawTileFloats = kTile * kYawObsWords + kTile;
constexpr int kRsYawTile = kRsNoise, kRsYawXch = kRsYawTile + 4 * kRsYawTileFloats, kRsYawSum = kRsYawXch + 4 * 256;
constexpr int kYawRow = 2 * kYawObsWords + 4;
static_assert((kRsYawXch * 4) % 8 == 0 && (kRsYawSum * 4) % 8 == 0 && kRsYawSum + 2 * 4 * kYawRow <= 2 * kNetFloats, "yaw phase-2 map");

struct YawTileRegs {
    Drone<float> d;
    YawExtra<float> e;
    uint64_t agent_tick;
    uint32_t led_fill;
};

__device__ __forceinline__ void rs_load_tile_yaw(const StepArgs<float>& s, YawTileRegs& tr) {
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (wave & 3);
    tr.agent_tick = s.tick[tile];
    tr.led_fill = 0u;
    if (wave < 4) {
        load_drone<float, kYawGranules>(tr.d, s.state, tile, lane);
        const Vec4<float>* ext = s.state + tile * (kYawGranules * kTile) + lane;
        const Vec4<float> g9 = ext[9 * kTile], g10 = ext[10 * kTile];
        tr.e = YawExtra<float>{g9.x, g9.y, g9.z, g9.w, g10.x, g10.y, g10.z, g10.w};
        tr.led_fill = s.led_count[tile];
    }
}

// Called by all 512 threads after the clipped actions are in LDS and a workgroup barrier.  Two workgroup barriers on every path.
__device__ __forceinline__ void rs_tail_yaw(const StepArgs<float>& s, float* lds, YawTileRegs& tr, const TailOut& o_) {
    using T = float;
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    const int tl = wave & 3;
    const int64_t tile = (int64_t)blockIdx.x * 4 + tl;
    const int64_t i = tile * kTile + lane;
    const bool live = i < s.n;
    Vec4<T>* tl_lds = reinterpret_cast<Vec4<T>*>(lds + kRsYawTile + tl * kRsYawTileFloats);
    double* bsum = reinterpret_cast<double*>(lds + kRsYawSum);
    if (wave >= 4) {
        lds_barrier();   // A: the physics wave has staged the observation tile (and the returns)
        const int64_t rows_left = s.n - tile * kTile;
        if (o_.partials) {
            const int rows = (int)(rows_left < kTile ? (rows_left > 0 ? rows_left : 0) : kTile);
            tile_moment_sums<kYawObsWords>(reinterpret_cast<const float*>(tl_lds), reinterpret_cast<double*>(lds + kRsYawXch + tl * 256),
                                           rows, lane, bsum + tl * kYawRow);
        }
        if (o_.store_obs) {
            Vec4<T>* out = reinterpret_cast<Vec4<T>*>(s.obs + tile * (kTile * kYawObsWords));
#pragma unroll
            for (int r = 0; r < kYawObsWords / 4; ++r) {
                const int f4 = r * kTile + lane;
                if (f4 / (kYawObsWords / 4) < rows_left) out[f4] = tl_lds[f4];
            }
        }
        lds_barrier();   // B: the four tiles' sums are in LDS
        if (o_.partials && wave == 4 && lane < kYawRow)
            o_.partials[(int64_t)blockIdx.x * kYawRow + lane] =
                (bsum[lane] + bsum[kYawRow + lane]) + (bsum[2 * kYawRow + lane] + bsum[3 * kYawRow + lane]);
        return;
    }
    // ---------------- physics wave: yaw_step_kernel's body
    Drone<T>& d = tr.d;
    YawExtra<T>& e = tr.e;
    const T a = live ? lds[kRsAct + 4 * (tl * kTile + lane)] : T(0);
    T ret_in = T(0);
    if (s.ret && live) ret_in = s.ret[i];
    const uint32_t gid = (uint32_t)(s.id_offset + (uint64_t)i);
    const T prev_a = e.h3;
    e.h0 = e.h1; e.h1 = e.h2; e.h2 = e.h3; e.h3 = a;                   // yaw.py:107
    T pwm[4];
    control_tick(d, T(0), T(0), a * T(-30), T(0), pwm);                // yaw.py:110-121
    T ga[4], gb[4];
    normal8(s.seed, gid, tr.agent_tick, ga, gb);
    const T inv_mass = rcp_(e.mass), k_lin = T(0.5 * 1.225 * 3.0e-4) * e.cd;
    physics_substep<T, true>(d, pwm, s.sigma, ga, inv_mass, k_lin, e.kpqr);
    physics_substep<T, true>(d, pwm, s.sigma, gb, inv_mass, k_lin, e.kpqr);
    T roll, pitch, yaw;
    euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
    uint32_t bits = 0;
    if (d.steps >= T(400)) bits |= kInfoTrunc;                          // yaw.py:138
    if (sqrt_(d.px * d.px + d.py * d.py + d.pz * d.pz) > T(3.0)) bits |= kInfoTerm | kInfoOob;  // yaw.py:141-143
    d.steps += T(1);                                                    // yaw.py:145
    const T av0 = clip_(wrap_angle(roll - d.er) * T(4), T(-1), T(1));
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
            const int64_t row = (int64_t)tr.led_fill + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (row < s.led_rows) {
                const int64_t slot = tile * s.led_rows + row;
                s.led_idx[slot] = (int32_t)i;
                s.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                s.led_tag[slot] = (uint32_t)tr.agent_tick;
                s.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(s.led_obs + slot * kYawObsWords);
#pragma unroll
                for (int g = 0; g < kYawObsWords / 4; ++g) lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            yaw_reset(d, e, warm_throttle<T>(), s.seed, gid, tr.agent_tick, s.domain_rand != 0, o);
        }
        tr.led_fill += (uint32_t)__popcll(m);
        if (lane == 0) s.led_count[tile] = tr.led_fill;
    }
#pragma unroll
    for (int g = 0; g < kYawObsWords / 4; ++g)
        tl_lds[lane * (kYawObsWords / 4) + g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
    {
        T ret = T(0);
        if (s.ret) {
            ret = live ? ret_in * (T)s.gamma + reward : T(0);
            if (live) s.ret[i] = done ? T(0) : ret;
        }
        reinterpret_cast<T*>(tl_lds)[kTile * kYawObsWords + lane] = ret;
        if (live && o_.done) o_.done[i] = done ? 1 : 0;
    }
    lds_barrier();   // A
    store_drone<T, kYawGranules>(d, s.state, tile, lane);
    Vec4<T>* ext = s.state + tile * (kYawGranules * kTile) + lane;
    ext[9 * kTile] = Vec4<T>{e.h0, e.h1, e.h2, e.h3};
    ext[10 * kTile] = Vec4<T>{e.mass, e.cd, e.kpqr, e.spare};
    if (lane == 0) s.tick[tile] = tr.agent_tick + 1ull;
    if (live) {
        o_.reward[i] = reward;
        o_.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        o_.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
    lds_barrier();   // B
}

// ---------------------------------------------------------------- the one-step kernel (TASK 0: hover, 1: yaw)
template <int TASK>
__global__ __launch_bounds__(512) void rollout_step_kernel(RollStepArgs p) {
    constexpr int IN = TASK == 0 ? kIn : kYawObsWords, ACT = TASK == 0 ? kAct : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
#define RS1_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(int64_t)blockIdx.x * kStampStride + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RS1_STAMP(0);
    f32x2 xpre[IN / 4];
    rollstep_prologue<IN>(p, lds, xpre);
    RS1_STAMP(1);
    // the drone state is requested here: it is needed only in phase 2, so its latency hides under the matrix work
    TileRegs tr;
    YawTileRegs ytr;
    if constexpr (TASK == 0) rs_load_tile(p.st, tr);
    else rs_load_tile_yaw(p.st, ytr);
    f32x4 cl;
    {
        const int64_t d = (int64_t)blockIdx.x * 256 + (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
        // the agent tick is this workgroup's own tile's copy (all copies are equal, and tile 0's is advanced by workgroup 0
        // in phase 2 while other workgroups may still be here)
        const PolicyOut out{p.obs_n, p.action, p.clipped, p.value, p.logp};
        cl = rs_policy_core<IN, ACT>(lds, xpre, d, d < p.st.n, p.rms != nullptr, p.clip_obs, out,
                                     p.st.tick[(int64_t)blockIdx.x * 4], p.pol_seed, p.pol_id_offset, (int)(threadIdx.x & 63));
    }
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    lds_barrier();                                     // every wave is done with the weight image: its space is free
    RS1_STAMP(2);
    if (lane < 32) reinterpret_cast<f32x4*>(lds + kRsAct)[wave * 32 + lane] = cl;
    lds_barrier();
    const TailOut to{p.st.reward, p.st.terminated, p.st.truncated, p.st.done, p.st.partials, true, false, true,
                     p.stamps ? p.stamps + (int64_t)blockIdx.x * kStampStride + 4 : nullptr};
    if constexpr (TASK == 0) {
        rs_tail(p.st, lds, tr, to);              // (stores the state itself: TailOut.store_state)
    } else {
        rs_tail_yaw(p.st, lds, ytr, to);
    }
    RS1_STAMP(3);
#undef RS1_STAMP
}

// ---------------------------------------------------------------- phase 2 alone: the env step of fpv_step / fpv_step_fused
// The same four-tiles-per-workgroup step (physics wave + helper wave per tile: noise hand-off, shared pixel-detector rows, tile
// store and moment sums beside the ledger writes) as a launch of its own, for actions that come from memory.  Same bits as
// step_kernel (fpv_step.hip) — rs_tail is what tests/test_rollstep_gpu.py compares with it — in less time where the chip would
// otherwise sit at one wave per SIMD with the pixel detector's row walk in that one wave (launch_step's dispatch, fpv_step.hip).
__global__ __launch_bounds__(512) void step_quad_kernel(StepArgs<float> st, uint64_t* stamps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TileRegs tr;
    rs_load_tile(st, tr);
    if (threadIdx.x < 256) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        reinterpret_cast<f32x4*>(lds + kRsAct)[threadIdx.x] = i < st.n ? reinterpret_cast<const f32x4*>(st.actions)[i] : f32x4{0.f, 0.f, 0.f, -1.f};
    }
    __syncthreads();
    const TailOut to{st.reward, st.terminated, st.truncated, st.done, st.partials, true, false, true,
                     stamps ? stamps + (int64_t)blockIdx.x * kStampStride + 4 : nullptr};
    rs_tail(st, lds, tr, to);
}

hipError_t launch_step_quad(const StepArgs<float>& st, uint64_t* stamps, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    constexpr int kPixelBytes = (kRsPix + RsPix::kWords) * 4, kAnalyticBytes = kRsPix * 4;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)step_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPixelBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int64_t tiles = (st.n + kTile - 1) / kTile;
    hipLaunchKernelGGL(step_quad_kernel, dim3((unsigned)(tiles / 4)), dim3(512), (size_t)(st.detector ? kPixelBytes : kAnalyticBytes), s,
                       st, stamps);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the one-step kernel with the policy on the 6-term bf16 split
// rollout_step_kernel<0> with fpv_policy_split.hip's arithmetic in phase 1 (hover task): the split image of one net is 121 KB, so
// the workgroup stages pi, evaluates pi and samples for its 256 drones, stages V over it, evaluates V, and then steps the same
// drones exactly as above (rs_tail).  The policy pass already holds 238 registers, so the drone state is requested after it —
// the previous launch's stores left it in L2 / MALL — not under the matrix work.  Bit-identical to fpv_policy_forward in gemm
// mode 1 followed by fpv_step_fused (tests/test_rollstep_gpu.py).  Not the headline: the exact-fp32 policy is.
struct RollStepSplitArgs {
    const unsigned char* image;   // split image: two nets + heads (kSplitImageBytes)
    const double* rms;
    float *obs_n, *action, *clipped, *value, *logp;
    uint64_t pol_seed, pol_id_offset;
    float clip_obs, eps_norm;
    MergeArgs mg;
    StepArgs<float> st;
    uint64_t* stamps;
};
constexpr int kRsSplitLdsBytes = (kRsPix + RsPix::kWords) * 4 > kSplitLdsBytes ? (kRsPix + RsPix::kWords) * 4 : kSplitLdsBytes;

#pragma clang fp contract(fast)
// WITHV = false (fpv_policy_io.nets = 1, "the critic deferred"): pi only.  V(s_t) enters GAE alone, the normalised row is in the
// rollout buffer, and fpv_policy_forward evaluates V-only on stored rows to the same bits — so a collector can run V over all T x N
// rows once, behind the last step, with V's image staged once per workgroup instead of once per step and workgroup.  Measured and
// NOT the default: the step drops 57 -> 43 us, the batched V pass costs 12.3-12.7 us per 65 536 rows (a sustained bf16
// stream runs at the chip's lower clock), 0.7-1.4 us per step more than it returns.
template <int TASK, bool WITHV>                        // TASK 0: hover (20 -> 4), 1: yaw (12 -> 1; round 5)
__global__ __launch_bounds__(512) void rollout_step_split_kernel(RollStepSplitArgs p) {
    constexpr int IN = TASK == 0 ? kIn : kYawObsWords, ACT = TASK == 0 ? kAct : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char* lds_raw = reinterpret_cast<unsigned char*>(lds);
    float* head = reinterpret_cast<float*>(lds_raw + kNetBytes);            // kHeadFloats, then mean[20], inv[20]
    float* nrm = head + kHeadFloats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31;
    const float clip = p.rms ? p.clip_obs : 3.0e38f;
#define RSS_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(int64_t)blockIdx.x * kStampStride + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RSS_STAMP(0);
    for (int q = threadIdx.x; q < kHeadFloats; q += 512)
        head[q] = reinterpret_cast<const float*>(p.image + 2 * (size_t)kNetBytes)[q];
    // pi's image by LDS-DMA (FPV_RSS_DMA_PI): requested BEHIND the merge table's loads and in front of their sums, inside the same
    // branch, so that the two travel together — a load hipcc knows of is waited for with a counter threshold, which also waits for
    // every DMA issued before it (the head block above is therefore read first), and a load under a branch is waited for at the join
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    (void)lds0;
    if (p.mg.rows) {                            // fold the previous env step's tile sums into the statistics first
        MergeRegs mr;
        merge_request(p.mg, mr);
#if FPV_RSS_DMA_PI
        split_stage_dma<512>(p.image, 0, lds0);
#endif
        merge_prologue(p.mg, mr, p.rms, reinterpret_cast<double*>(nrm + 40), p.eps_norm, nrm, nrm + 20);
    } else {
#if FPV_RSS_DMA_PI
        split_stage_dma<512>(p.image, 0, lds0);
#endif
        if ((int)threadIdx.x < IN) {
            nrm[threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
            nrm[20 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[IN + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
        }
    }
    const int64_t d = (int64_t)blockIdx.x * 256 + wave * 32 + col;
    const uint64_t tick = p.st.tick[(int64_t)blockIdx.x * 4];       // this workgroup's own tile's copy (rollout_step_kernel)
    const SplitOut out{p.obs_n, p.action, p.clipped, p.value, p.logp, nullptr};
    f32x4 cl = f32x4{0.f, 0.f, 0.f, 0.f};
    // Each net's whole image in front of the net.  With REGISTER staging (FPV_RSS_DMA_PI / _V = 0) these were measured and not kept
    // (DESIGN §5b): pi's image requested before the statistics merge (+0.4 us per step), V's requested before pi's head arithmetic
    // (same), and the image in two parts — W1 + biases first, layer 1 on it while W2 is on its way — (+1.8 us: two more workgroup
    // barriers and 48 registers in flight cost more than the fetch they hide).  By LDS-DMA, which holds no registers, the first two
    // are what the kernel does now (-0.5 us together).
    SplitHidden hd;
    {
        SplitStage<512> stage;
#if FPV_RSS_DMA_PI
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        split_stage_request(p.image, 0, stage);
        split_stage_commit(stage, lds_raw);
#endif
        __syncthreads();                               // pi's image, the head block and the normaliser are in LDS
        RSS_STAMP(1);
        split_net_hidden<IN>(lds_raw, nrm, clip, p.st.obs, d, d < p.st.n, lane, p.obs_n, hd);
#if FPV_RSS_DMA_V
        // V's image by LDS-DMA over pi's as soon as every wave is through pi's layer 2, in flight under pi's head arithmetic and
        // sampling (which read the head block and registers only); the wait stands in front of the head's row stores
        if constexpr (WITHV) {
            lds_barrier();                             // every wave is done with pi's image
            split_stage_dma<512>(p.image, 1, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_raw);
            cl = split_net_head<ACT>(0, head, hd, d, d < p.st.n, lane, out, nullptr, tick, false, p.pol_seed, p.pol_id_offset,
                                     [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); });
            lds_barrier();                             // every wave's pieces have landed
        } else
#endif
        cl = split_net_head<ACT>(0, head, hd, d, d < p.st.n, lane, out, nullptr, tick, false, p.pol_seed, p.pol_id_offset);
        if constexpr (WITHV) {
#if !FPV_RSS_DMA_V
            __syncthreads();                           // every wave is done with pi's image
            split_stage_request(p.image, 1, stage);
            split_stage_commit(stage, lds_raw);
            __syncthreads();
#endif
            split_net_hidden<IN>(lds_raw, nrm, clip, p.st.obs, d, d < p.st.n, lane, nullptr, hd);
            (void)split_net_head<ACT>(1, head, hd, d, d < p.st.n, lane, out, nullptr, tick, false, p.pol_seed, p.pol_id_offset);
        }
    }
    TileRegs tr;
    YawTileRegs ytr;
    if constexpr (TASK == 0) rs_load_tile(p.st, tr);
    else rs_load_tile_yaw(p.st, ytr);
    lds_barrier();                                     // every wave is done with the image: its space is free
    RSS_STAMP(2);
    if (lane < 32) reinterpret_cast<f32x4*>(lds + kRsAct)[wave * 32 + lane] = cl;
    lds_barrier();
    const TailOut to{p.st.reward, p.st.terminated, p.st.truncated, p.st.done, p.st.partials, true, false, true,
                     p.stamps ? p.stamps + (int64_t)blockIdx.x * kStampStride + 4 : nullptr};
    if constexpr (TASK == 0) rs_tail(p.st, lds, tr, to);
    else rs_tail_yaw(p.st, lds, ytr, to);
    RSS_STAMP(3);
#undef RSS_STAMP
}
#pragma clang fp contract(on)

#ifdef FPV_EXPERIMENTS   // measured slower than one launch per step: built into libfpv_env_experiments.so only
// ---------------------------------------------------------------- the persistent kernel: T rollout steps in one launch
// One workgroup per CU keeps the weight image in LDS and its 256 drones' state in registers for the whole rollout.  What
// crosses workgroups is VecNormalize's statistics only: RunningMeanStd.update needs the moment sums of ALL drones of step t
// before any drone's observation of step t + 1 is normalised (SB3 VecNormalize.step_wait), so the steps are separated by a
// grid-wide barrier behind which every workgroup folds the same n / 256 rows in the same fixed order (the order of
// merge_prologue, fpv_internal.h: identical bits) and carries the running statistics in registers.  A kernel boundary costs
// the same wait plus a relaunch with cold weights; here the barrier replaces it.
//   hand-off (cdna_hip_programming.md Guideline 16, R1): rows leave with sc1 (write-through) stores, drained by the storing
//   wave, then a workgroup barrier, then thread 0 arrives: group counter (blockIdx & 7, one 128-B line each) -> top counter ->
//   the last arriver writes every group's generation word; thread 0 polls its group's word (relaxed sc1 load + s_sleep);
//   after a workgroup barrier all threads read the rows with sc1 loads.  Counters are monotonic over the launch and zeroed by
//   the host before it.
//   Every wait is bounded (s_memrealtime): on a timeout the waiter raises the abort word, every workgroup leaves its loop at
//   its next wait, and the status word tells the host that the rollout is invalid.  All workgroups must be co-resident: the
//   grid is at most one workgroup per CU (160 KB of LDS each) and the launch is a cooperative one, which checks that.
// Per step the first kRsEnd floats of the image (the phase-2 exchange space) are staged again from L2 while the barrier is
// awaited.
constexpr int kRsEnd = kRsSum + 2 * 4 * kPartialWords;      // 11 872 floats = 47.5 KB (the analytic detector's phase-2 map)
constexpr int kSyncStride = 32;                            // words: one 128-B line per counter
constexpr int kSyncGroups = kPrologueMergeWays;            // 9 groups = the 9 ways of merge_prologue's summation order
constexpr int kSyncTop = kSyncGroups, kSyncGen = kSyncGroups + 1, kSyncAbort = 2 * kSyncGroups + 1;
constexpr int kSyncCounterWords = (2 * kSyncGroups + 2) * kSyncStride;
constexpr int kSyncWayDoubles = kSyncGroups * kPartialWords;                     // the groups' sums, then the total
constexpr int kSyncWords = kSyncCounterWords + 2 * (kSyncWayDoubles + kPartialWords);

struct RollPersistArgs {
    const float* image;
    uint64_t pol_seed, pol_id_offset;
    float clip_obs, eps_norm;
    StepArgs<float> st;          // state / tick / ledger / ret / gamma / noise; st.obs: obs_0 in, obs_T out; st.terminated / truncated:
                                 // the last step's flags; st.reward / done / partials are not used
    int32_t T;
    float* obs_n;                // time-major rollout buffers, step t at base + t * n * width; obs_n / action / logp may be null
    float* action;
    float* value;
    float* logp;
    float* reward;               // [T][n] raw rewards
    uint8_t* done;               // [T][n] or null
    const double* rms0;          // [41] statistics entering step 0, or null (no observation normalisation)
    const double* ret_rms0;      // [3] or null
    double* rms_out[2];          // statistics after folding step t's sums -> [(t + 1) & 1]   (t = 0 .. T - 2)
    double* ret_rms_out[2];
    double* snapshots;           // row t <- statistics after folding step t's sums (t = 0 .. T - 2); may be null
    int64_t snap_stride;
    double* rows[2];             // [n / 256][kPartialWords]; step t's sums go to rows[t & 1]
    uint32_t* sync;              // kSyncWords words, zero at launch
    uint32_t* status;            // set to 1 when a grid wait timed out
    uint64_t timeout_ticks;      // of s_memrealtime (100 MHz)
    uint64_t* stamps;            // diagnostic (tools/rollout_phases.py): [grid][T][8] s_memrealtime values, or null
};

__device__ __forceinline__ uint32_t sync_load(const uint32_t* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The grid-wide barrier that carries the reduction.  Called by ALL threads after the workgroup's row of step sums has been
// stored write-through and drained and a workgroup barrier has passed.  On success xch[0 .. 43] holds the column sums of
// all rows, added in merge_prologue's order: way g = rows g, g + 9, g + 18, ... in that order, then the ways 0 .. 8 in order.
//   arrive on group counter g = blockIdx % 9; the group's last arriver adds the group's rows (that IS way g) and publishes the
//   way sum; it then arrives on the top counter, whose last arriver adds the nine way sums, publishes the total and releases
//   every group's generation word; everyone polls its group's word and reads the 352-byte total.
// All cross-workgroup data moves with sc1 stores (drained before the next counter operation) and sc1 loads; counters are
// monotonic over the launch.  epoch counts from 1.  flag: an LDS word.  Returns false when the wait was abandoned.
__device__ __forceinline__ bool grid_reduce(uint32_t* w, const double* rows, uint32_t epoch, uint32_t nblocks, uint64_t timeout_ticks,
                                            double* xch, int* flag) {
    const int tid = threadIdx.x;
    const uint32_t g = blockIdx.x % (uint32_t)kSyncGroups;
    const uint32_t members = (nblocks - g + kSyncGroups - 1) / kSyncGroups;      // workgroups b < nblocks with b % 9 == g
    const uint32_t groups = nblocks < (uint32_t)kSyncGroups ? nblocks : (uint32_t)kSyncGroups;
    double* way = reinterpret_cast<double*>(w + kSyncCounterWords);
    double* total = way + kSyncWayDoubles;
    if (tid == 0) {
        const uint32_t a = __hip_atomic_fetch_add(w + g * kSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = a + 1u == members * epoch;
    }
    lds_barrier();
    const bool group_last = *flag != 0;
    lds_barrier();
    if (group_last) {                                  // uniform
        if (tid < kPartialWords) {
            constexpr int kQ = (256 + kSyncGroups - 1) / kSyncGroups;             // 29: the grid has <= 256 rows
            double v[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t b = g + (uint32_t)q * kSyncGroups;
                v[q] = b < nblocks ? row_load_sc1(rows + (int64_t)b * kPartialWords + tid) : 0.0;
            }
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < kQ; ++q) acc += v[q];
            row_store_sc1(way + g * kPartialWords + tid, acc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
        if (tid == 0) {
            const uint32_t b = __hip_atomic_fetch_add(w + kSyncTop * kSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = b + 1u == groups * epoch;
        }
        lds_barrier();
        const bool top_last = *flag != 0;
        lds_barrier();
        if (top_last) {                                // uniform
            if (tid < kPartialWords) {
                double v[kSyncGroups];
#pragma unroll
                for (int k = 0; k < kSyncGroups; ++k)
                    v[k] = (uint32_t)k < groups ? row_load_sc1(way + k * kPartialWords + tid) : 0.0;
                double tt = 0.0;
#pragma unroll
                for (int k = 0; k < kSyncGroups; ++k) tt += v[k];
                row_store_sc1(total + tid, tt);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();
            if (tid == 0)
                for (uint32_t k = 0; k < groups; ++k)
                    __hip_atomic_store(w + (kSyncGen + k) * kSyncStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        bool ok = true;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t polls = 0;
        while (sync_load(w + (kSyncGen + g) * kSyncStride) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if ((++polls & 15u) == 1u) {                // on the first poll and every sixteenth after it
                if (sync_load(w + kSyncAbort * kSyncStride) != 0u) { ok = false; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                    __hip_atomic_store(w + kSyncAbort * kSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        *flag = ok ? 1 : 0;
    }
    lds_barrier();
    const bool ok = *flag != 0;
    if (ok && tid < kPartialWords) xch[tid] = row_load_sc1(total + tid);
    lds_barrier();
    return ok;
}

__global__ __launch_bounds__(512) void rollout_persistent_kernel(RollPersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hb = lds + 2 * kNetFloats;
    double* xch = reinterpret_cast<double*>(hb + 700);
    int* flag = reinterpret_cast<int*>(hb + 658);              // padding word of the head block: never read by the policy
    const StepArgs<float>& s = p.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
    const int64_t n = s.n;
    const int64_t d = (int64_t)blockIdx.x * 256 + wave * 32 + col;
    const bool normalise = p.rms0 != nullptr;

    f32x2 xraw[5];
    {
        const f32x2* row = reinterpret_cast<const f32x2*>(s.obs + d * kIn + kHalf * h);
#pragma unroll
        for (int q = 0; q < kHalf / 2; ++q) xraw[q] = row[q];
    }
    rs_stage<(kImageFloats / 4 + 511) / 512>(p.image, lds, 0, kImageFloats);
    // running statistics in registers: thread c < 20 owns observation channel c, thread 40 the discounted-return statistics
    const bool own_obs = tid < kIn && normalise, own_ret = tid == 2 * kIn && p.ret_rms0 != nullptr;
    double st_mean = 0.0, st_var = 1.0, st_cnt = 0.0;
    if (own_obs) { st_mean = p.rms0[tid]; st_var = p.rms0[kIn + tid]; st_cnt = p.rms0[2 * kIn]; }
    if (own_ret) { st_mean = p.ret_rms0[0]; st_var = p.ret_rms0[1]; st_cnt = p.ret_rms0[2]; }
    if (tid < kIn) {
        hb[660 + tid] = normalise ? (float)st_mean : 0.0f;
        hb[680 + tid] = normalise ? (float)(1.0 / sqrt(st_var + (double)p.eps_norm)) : 1.0f;
    }
    // The drone state is NOT carried in registers across the policy pass (it would push that pass over its register budget):
    // the physics waves store it at the end of their env step and ask for it again as soon as their policy pass is done —
    // they are the older waves of their SIMDs and finish it ~28 k cycles before the younger ones, so the loads have landed
    // by the time the workgroup meets.  The tick stays in a register (it is read by the helper waves too).
    TileRegs tr;
    tr.agent_tick = s.tick[(int64_t)blockIdx.x * 4 + (wave & 3)];
    const uint64_t pol_tick0 = s.tick[(int64_t)blockIdx.x * 4];
    const double bc = (double)n;
    const int nrows = (int)gridDim.x;
    bool ok = true;
    lds_barrier();

#define RS_STAMP(k) do { if (p.stamps && tid == 0) p.stamps[((int64_t)blockIdx.x * p.T + t) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#pragma unroll 1
    for (int t = 0; t < p.T; ++t) {
        RS_STAMP(0);
        const int64_t off = (int64_t)t * n;
        const PolicyOut out{p.obs_n ? p.obs_n + off * kIn : nullptr, p.action ? p.action + off * kAct : nullptr, nullptr,
                            p.value + off, p.logp ? p.logp + off : nullptr};
        // opaque per iteration: hipcc otherwise hoists the layers' LDS addresses and the head weights' LDS loads (hundreds of
        // registers) out of the step loop
        int lane_o = lane, zero = 0;
        asm volatile("" : "+v"(lane_o), "+s"(zero));
        const f32x4 cl = rs_policy_core<kIn, kAct>(lds + zero, xraw, d, true, normalise, p.clip_obs, out, pol_tick0 + (uint64_t)t, p.pol_seed,
                                        p.pol_id_offset, lane_o);
        rs_load_state(s, tr);
        lds_barrier();                               // every wave is done with the weight image's head: phase-2 space is free
        RS_STAMP(1);
        if (lane < 32) reinterpret_cast<f32x4*>(lds + kRsAct)[wave * 32 + lane] = cl;
        lds_barrier();
        const bool last = t == p.T - 1;
        const TailOut to{p.reward + off, s.terminated, s.truncated, p.done ? p.done + off : nullptr, p.rows[t & 1], last, !last, false, nullptr};
        rs_tail(s, lds, tr, to);
        rs_store_state(s, tr);
        RS_STAMP(2);
        if (last) break;

        // ---- hand-over to step t + 1
        {   // this lane's raw observation words for the next policy pass, from the staged tiles
            const float* tb = lds + kRsTile + (wave >> 1) * kRsTileFloats + ((wave & 1) * 32 + col) * kObsWords + kHalf * h;
#pragma unroll
            for (int q = 0; q < kHalf / 2; ++q) xraw[q] = *reinterpret_cast<const f32x2*>(tb + 2 * q);
        }
        {   // the weight image's head again: requested now, written once every wave has read its rows
            constexpr int kGran = kRsEnd / 4, kPer = (kGran + 511) / 512;
            const f32x4* src = reinterpret_cast<const f32x4*>(p.image);
            f32x4 stage[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * 512 + tid;
                stage[q] = src[g < kGran ? g : 0];
            }
            lds_barrier();                           // rows read; the row's write-through stores are drained (rs_tail)
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * 512 + tid;
                if (g < kGran) reinterpret_cast<f32x4*>(lds)[g] = stage[q];
            }
        }
        RS_STAMP(3);
        ok = grid_reduce(p.sync, p.rows[t & 1], (uint32_t)(t + 1), (uint32_t)nrows, p.timeout_ticks, xch, flag);
        RS_STAMP(4);
        if (!ok) break;                                // uniform

        // ---- RunningMeanStd.update with step t's column sums, every workgroup alike (merge_prologue's expressions)
        {
            const bool publish = blockIdx.x == 0;
            if (own_obs || own_ret) {
                const double bm = (own_obs ? xch[tid] : xch[2 * kIn]) / bc;
                double bv = (own_obs ? xch[kIn + tid] : xch[2 * kIn + 1]) / bc - bm * bm;
                if (bv < 0.0) bv = 0.0;
                const double delta = bm - st_mean, tot = st_cnt + bc;
                const double nm = st_mean + delta * bc / tot;
                const double nv = (st_var * st_cnt + bv * bc + delta * delta * st_cnt * bc / tot) / tot;
                st_mean = nm; st_var = nv; st_cnt = tot;
                double* snap = p.snapshots ? p.snapshots + (int64_t)t * p.snap_stride : nullptr;
                if (own_obs) {
                    hb[660 + tid] = (float)nm;
                    hb[680 + tid] = (float)(1.0 / sqrt(nv + (double)p.eps_norm));
                    if (publish) {
                        double* ro = p.rms_out[(t + 1) & 1];
                        if (ro) { ro[tid] = nm; ro[kIn + tid] = nv; if (tid == 0) ro[2 * kIn] = tot; }
                        if (snap) { snap[tid] = nm; snap[kIn + tid] = nv; if (tid == 0) snap[2 * kIn] = tot; }
                    }
                } else if (publish) {
                    double* ro = p.ret_rms_out[(t + 1) & 1];
                    if (ro) { ro[0] = nm; ro[1] = nv; ro[2] = tot; }
                    if (snap) { snap[41] = nm; snap[42] = nv; snap[43] = tot; }
                }
            }
            lds_barrier();
        }
        RS_STAMP(5);
    }
#undef RS_STAMP
    if (wave < 4 && lane == 0) s.tick[(int64_t)blockIdx.x * 4 + (wave & 3)] = tr.agent_tick;
    if (!ok && tid == 0) *p.status = 1u;
}

#endif   // FPV_EXPERIMENTS

// ---------------------------------------------------------------- launchers
// Preconditions (checked by fpv_rollout_step): fp32 handle with motor noise, n a multiple of 256, the task's policy widths.
hipError_t launch_rollout_step(const float* image, const double* rms, float* obs_n, float* action, float* clipped, float* value,
                               float* logp, uint64_t pol_seed, uint64_t pol_id_offset, float clip_obs, float eps_norm,
                               const MergeArgs& merge, const StepArgs<float>& st, int yaw, uint64_t* stamps, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    const int full = (int)((2 * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float)) + kMergeLdsBytes;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rollout_step_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)rollout_step_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    RollStepArgs p{image, rms, obs_n, action, clipped, value, logp, pol_seed, pol_id_offset, clip_obs, eps_norm, merge, st, stamps};
    if (yaw) hipLaunchKernelGGL(rollout_step_kernel<1>, dim3((unsigned)(st.n / 256)), dim3(512), (size_t)full, s, p);
    else hipLaunchKernelGGL(rollout_step_kernel<0>, dim3((unsigned)(st.n / 256)), dim3(512), (size_t)full, s, p);
    return hipGetLastError();
}

hipError_t launch_rollout_step_split(const unsigned char* image, const double* rms, float* obs_n, float* action, float* clipped, float* value,
                                     float* logp, uint64_t pol_seed, uint64_t pol_id_offset, float clip_obs, float eps_norm,
                                     const MergeArgs& merge, const StepArgs<float>& st, int yaw, int with_v, uint64_t* stamps, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    if (!attr_set) {
        const void* const kernels[4] = {(const void*)rollout_step_split_kernel<0, true>, (const void*)rollout_step_split_kernel<1, true>,
                                        (const void*)rollout_step_split_kernel<0, false>, (const void*)rollout_step_split_kernel<1, false>};
        for (const void* k : kernels) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kRsSplitLdsBytes);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    RollStepSplitArgs p{image, rms, obs_n, action, clipped, value, logp, pol_seed, pol_id_offset, clip_obs, eps_norm, merge, st, stamps};
    const dim3 grid((unsigned)(st.n / 256));
    if (with_v) {
        if (yaw) hipLaunchKernelGGL((rollout_step_split_kernel<1, true>), grid, dim3(512), (size_t)kRsSplitLdsBytes, s, p);
        else hipLaunchKernelGGL((rollout_step_split_kernel<0, true>), grid, dim3(512), (size_t)kRsSplitLdsBytes, s, p);
    } else {
        if (yaw) hipLaunchKernelGGL((rollout_step_split_kernel<1, false>), grid, dim3(512), (size_t)kRsSplitLdsBytes, s, p);
        else hipLaunchKernelGGL((rollout_step_split_kernel<0, false>), grid, dim3(512), (size_t)kRsSplitLdsBytes, s, p);
    }
    return hipGetLastError();
}

#ifdef FPV_EXPERIMENTS
int rollout_sync_words() { return kSyncWords; }

hipError_t launch_rollout_persistent(const RollPersistHost& h, const StepArgs<float>& st, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    const int full = (int)((2 * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float)) + kMergeLdsBytes;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rollout_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    RollPersistArgs p{};
    p.image = h.image; p.pol_seed = h.pol_seed; p.pol_id_offset = h.pol_id_offset; p.clip_obs = h.clip_obs; p.eps_norm = h.eps_norm;
    p.st = st; p.T = h.T;
    p.obs_n = h.obs_n; p.action = h.action; p.value = h.value; p.logp = h.logp; p.reward = h.reward; p.done = h.done;
    p.rms0 = h.rms0; p.ret_rms0 = h.ret_rms0;
    p.rms_out[0] = h.rms_out[0]; p.rms_out[1] = h.rms_out[1];
    p.ret_rms_out[0] = h.ret_rms_out[0]; p.ret_rms_out[1] = h.ret_rms_out[1];
    p.snapshots = h.snapshots; p.snap_stride = h.snap_stride;
    p.rows[0] = h.rows[0]; p.rows[1] = h.rows[1];
    p.sync = h.sync; p.status = h.status; p.timeout_ticks = h.timeout_ticks;
    p.stamps = h.stamps;
    // Zeroed by a kernel, not hipMemsetAsync: a memset node captured into a HIP graph is right on the graph's first replay only —
    // from the second replay on it may fill the words with stale kernel-argument bytes (device-address fragments) instead of the
    // value (tools/graph_memset_probe.py, profiles/r03_graph_memset_probe.json, tests/test_graph_memset_gpu.py).  Round 2 issued the
    // memset here: the barrier words then held garbage from the second replay of a captured rollout on, every poll passed at
    // once and the steps read the previous rollout's column sums without any wait timing out.
    hipError_t e = launch_zero_words(h.sync, kSyncCounterWords, s);
    if (e != hipSuccess) return e;
    void* args[] = {&p};
    // cooperative launch: the runtime checks that the grid fits on the device at once (the barrier needs every workgroup resident)
    return hipLaunchCooperativeKernel((const void*)rollout_persistent_kernel, dim3((unsigned)(st.n / 256)), dim3(512), args,
                                      (unsigned)full, s);
}

#else
int rollout_sync_words() { return 0; }
#endif   // FPV_EXPERIMENTS

}  // namespace fpv
