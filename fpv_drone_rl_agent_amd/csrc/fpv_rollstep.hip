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
// Reference path replaced: QuadXHoverEnv.step()/reset() simulation/hover.py:72-114,334-358 under SB3's VecEnv auto-reset,
// together with MlpPolicy.forward + VecNormalize of simulation/train_hover.py:43-58.
#include "fpv_mlp.h"
#include "fpv_policy_image.h"
#include "fpv_step_parts.h"

namespace fpv {

struct RollStepArgs {
    // ---- policy side (fpv_policy_io)
    const float* image;      // packed weights (kImageFloats), hover widths 20 / 4
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

// ---------------------------------------------------------------- phase 1: fpv_policy.hip's body for one 256-drone chunk
#pragma clang fp contract(fast)
__device__ __forceinline__ f32x4 rollstep_policy_phase(const RollStepArgs& p, float* lds) {
    constexpr int THREADS = 512;
    float* hb = lds + 2 * kNetFloats;                // heads [660], then normaliser mean[20], inv[20]
    const int64_t n = p.st.n;
    f32x4 xpre[3];
    {
        const int64_t d0 = (int64_t)blockIdx.x * (THREADS / 2) + (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
        const int hh = (threadIdx.x >> 5) & 1;
        const f32x4* row = reinterpret_cast<const f32x4*>(p.st.obs + (d0 < n ? d0 : 0) * kIn);
#pragma unroll
        for (int q = 0; q < 3; ++q) xpre[q] = row[4 * (3 * hh + q) < kIn ? 3 * hh + q : 0];
    }
    MergeRegs mr;
    if (p.mg.rows) merge_request(p.mg, mr);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.image);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        constexpr int kNetGran = kNetFloats / 4, kPer = (kNetGran + THREADS - 1) / THREADS;
        constexpr int kHeadGran = (kImageFloats - kHeadBase) / 4;
        const f32x4 hv = reinterpret_cast<const f32x4*>(p.image + kHeadBase)[threadIdx.x < kHeadGran ? threadIdx.x : 0];
        for (int sl = 0; sl < 2; ++sl) {
            f32x4 stage[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * THREADS + threadIdx.x;
                stage[q] = src[sl * kNetGran + (g < kNetGran ? g : 0)];
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * THREADS + threadIdx.x;
                if (g < kNetGran) dst[sl * kNetGran + g] = stage[q];
            }
        }
        if (threadIdx.x < kHeadGran) reinterpret_cast<f32x4*>(hb)[threadIdx.x] = hv;
        if (!p.mg.rows && (int)threadIdx.x < kIn) {
            hb[660 + threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
            hb[680 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[kIn + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
        }
    }
    if (p.mg.rows) merge_prologue(p.mg, mr, p.rms, reinterpret_cast<double*>(hb + 700), p.eps_norm, hb + 660, hb + 680);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const float clip = p.rms ? p.clip_obs : 3.0e38f;
    const int64_t d = (int64_t)blockIdx.x * (THREADS / 2) + wave * 32 + col;
    const bool live = d < n;
    float x[12];
    {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bool have = live && 4 * (3 * h + q) < kIn;
            f32x4 v = xpre[q];
            if (!have) v = f32x4{0.f, 0.f, 0.f, 0.f};
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int k = 12 * h + j;
            if (k < kIn) x[j] = fminf(fmaxf((x[j] - hb[660 + k]) * hb[680 + k], -clip), clip);
        }
        if (p.obs_n && live) {
            f32x4* out = reinterpret_cast<f32x4*>(p.obs_n + d * kIn);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (4 * (3 * h + q) < kIn) out[3 * h + q] = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
        }
    }
    float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int net = 0; net < 2; ++net) {
        const float* nb = lds + net * kNetFloats;
        Hidden a;
        bias_init(a, nb + kOffB1, h);
#pragma unroll
        for (int s = 0; s < 12; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[kOffW1 + (m * 12 + s) * 64 + lane], x[s], a.t[m], 0, 0, 0);
        tanh_all(a);
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
        tanh_all(c);
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
    if (live && h == 1) p.value[d] = head[4] + hb[644];
    const float mu[4] = {head[0] + hb[640], head[1] + hb[641], head[2] + hb[642], head[3] + hb[643]};

    // sample: Philox4x32-10 keyed like fpv_policy.hip (stream tag 'POL1'); the agent tick is this workgroup's own tile's copy
    // (all copies are equal, and tile 0's is advanced by workgroup 0 in phase 2 while others may still be here)
    float eps[4];
    {
        const uint64_t t = p.st.tick[(int64_t)blockIdx.x * 4];
        uint32_t xr[4];
        philox4x32_10((uint32_t)(p.pol_id_offset + (uint64_t)d), (uint32_t)t, (uint32_t)(t >> 32), 0x504F4C31u,
                      (uint32_t)p.pol_seed, (uint32_t)(p.pol_seed >> 32), xr);
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
        lp -= 0.5f * eps[j] * eps[j];
    }
    const f32x4 cl = f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f),
                           fminf(fmaxf(act[2], -1.f), 1.f), fminf(fmaxf(act[3], -1.f), 1.f)};
    if (live) {
        if (h == 0) {
            if (p.action) reinterpret_cast<f32x4*>(p.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
            if (p.clipped) reinterpret_cast<f32x4*>(p.clipped)[d] = cl;
        } else if (p.logp) {
            p.logp[d] = lp;
        }
    }
    return cl;
}
#pragma clang fp contract(on)

// ---------------------------------------------------------------- the kernel
__global__ __launch_bounds__(512) void rollout_step_kernel(RollStepArgs p) {
    using T = float;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const f32x4 cl = rollstep_policy_phase(p, lds);

    const StepArgs<float>& s = p.st;
    const int lane = threadIdx.x & (kTile - 1), wave = threadIdx.x >> 6;
    __syncthreads();                                   // every wave is done with the weight image: its space is free
    if (lane < 32) reinterpret_cast<f32x4*>(lds + kRsAct)[wave * 32 + lane] = cl;
    __syncthreads();

    const int tl = wave & 3;                           // tile of this workgroup served by this wave
    const bool helper = wave >= 4;
    const int64_t tile = (int64_t)blockIdx.x * 4 + tl;
    const int64_t i = tile * kTile + lane;
    const bool live = i < s.n;
    const uint64_t agent_tick = s.tick[tile];
    const uint32_t gid = (uint32_t)(s.id_offset + (uint64_t)i);
    T(*nz)[8][kTile] = reinterpret_cast<T(*)[8][kTile]>(lds + kRsNoise + tl * (2 * 8 * kTile));
    Vec4<T>* tl_lds = reinterpret_cast<Vec4<T>*>(lds + kRsTile + tl * kRsTileFloats);
    double* bsum = reinterpret_cast<double*>(lds + kRsSum);

    if (helper) {
        // ---- noise producer: pair k is published before barrier k (fpv_step.hip, step_kernel_pair)
#pragma unroll 1
        for (int k = 0; k < kSubstepPairs; ++k) {
            T ga[4], gb[4];
            normal8(s.seed, gid, agent_tick * (uint64_t)kSubstepPairs + (uint64_t)k, ga, gb);
            T(*slot)[kTile] = nz[k & 1];
            slot[0][lane] = ga[0]; slot[1][lane] = ga[1]; slot[2][lane] = ga[2]; slot[3][lane] = ga[3];
            slot[4][lane] = gb[0]; slot[5][lane] = gb[1]; slot[6][lane] = gb[2]; slot[7][lane] = gb[3];
            __syncthreads();
        }
        __syncthreads();   // barrier 6: the physics wave has staged the observation tile (and the returns)
        Vec4<T>* out = reinterpret_cast<Vec4<T>*>(s.obs + tile * (kTile * kObsWords));
        const int64_t rows_left = s.n - tile * kTile;
#pragma unroll
        for (int r = 0; r < kObsWords / 4; ++r) {
            const int f4 = r * kTile + lane;
            if (f4 / (kObsWords / 4) < rows_left) out[f4] = tl_lds[f4];
        }
        if (s.partials) {
            const int rows = (int)(rows_left < kTile ? (rows_left > 0 ? rows_left : 0) : kTile);
            tile_moment_sums<kObsWords>(reinterpret_cast<const float*>(tl_lds), reinterpret_cast<double*>(lds + kRsXch + tl * 256),
                                        rows, lane, bsum + tl * kPartialWords);
        }
        __syncthreads();   // barrier 7: the four tiles' sums are in LDS
        if (s.partials && wave == 4 && lane < kPartialWords)
            s.partials[(int64_t)blockIdx.x * kPartialWords + lane] =
                (bsum[lane] + bsum[kPartialWords + lane]) + (bsum[2 * kPartialWords + lane] + bsum[3 * kPartialWords + lane]);
        return;
    }

    // ---------------- physics wave
    Drone<T> d;
    load_drone(d, s.state, tile, lane);
    const f32x4 av = reinterpret_cast<const f32x4*>(lds + kRsAct)[tl * kTile + lane];
    const Vec4<T> act = live ? Vec4<T>{av.x, av.y, av.z, av.w} : Vec4<T>{T(0), T(0), T(0), T(-1)};
    const uint32_t led_fill = s.led_count[tile];
    const T tmpl_lane = s.tmpl[lane < kStateWords + kObsWords ? lane : 0];
    T ret_in = T(0);
    if (s.ret && live) ret_in = s.ret[i];
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
        physics_substep<T, true>(d, pwm, s.sigma, ga);
        physics_substep<T, true>(d, pwm, s.sigma, gb);
    }
    T roll, pitch, yaw;
    euler_from_quat(d.qx, d.qy, d.qz, d.qw, roll, pitch, yaw);
    T o[kObsWords];
    const bool vis = observe(d, roll, pitch, yaw, act.x, act.y, act.z, act.w, o, s.detector);
    uint32_t bits;
    const T reward = reward_and_flags(d, roll, pitch, vis, o, act.x, act.y, act.z, act.w, s.floor_check != 0, bits);
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
            if (row < s.led_rows) {
                const int64_t slot = tile * s.led_rows + row;
                s.led_idx[slot] = (int32_t)i;
                s.led_info[slot] = bits | ((uint32_t)d.steps << 8);
                s.led_tag[slot] = (uint32_t)agent_tick;
                s.led_ret[slot] = d.ep_ret;
                Vec4<T>* lo = reinterpret_cast<Vec4<T>*>(s.led_obs + slot * kObsWords);
#pragma unroll
                for (int g = 0; g < kObsWords / 4; ++g) lo[g] = Vec4<T>{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
            }
            d = r;
#pragma unroll
            for (int j = 0; j < kObsWords; ++j) o[j] = ro[j];
        }
        if (lane == 0) s.led_count[tile] = led_fill + (uint32_t)__popcll(m);
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
        if (live && s.done) s.done[i] = done ? 1 : 0;
    }
    __syncthreads();   // barrier 6
    store_drone(d, s.state, tile, lane);
    if (lane == 0) s.tick[tile] = agent_tick + 1ull;
    if (live) {
        s.reward[i] = reward;
        s.terminated[i] = (bits & kInfoTerm) ? 1 : 0;
        s.truncated[i] = (bits & kInfoTrunc) ? 1 : 0;
    }
    __syncthreads();   // barrier 7
}

// ---------------------------------------------------------------- launcher
// Preconditions (checked by fpv_rollout_step): fp32 hover handle with motor noise, n a multiple of 256, policy widths 20 / 4.
hipError_t launch_rollout_step(const float* image, const double* rms, float* obs_n, float* action, float* clipped, float* value,
                               float* logp, uint64_t pol_seed, uint64_t pol_id_offset, float clip_obs, float eps_norm,
                               const MergeArgs& merge, const StepArgs<float>& st, hipStream_t s) {
    static bool attr_set = false;
    const int full = (int)((2 * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float)) + kMergeLdsBytes;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rollout_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    RollStepArgs p{image, rms, obs_n, action, clipped, value, logp, pol_seed, pol_id_offset, clip_obs, eps_norm, merge, st};
    hipLaunchKernelGGL(rollout_step_kernel, dim3((unsigned)(st.n / 256)), dim3(512), (size_t)full, s, p);
    return hipGetLastError();
}

}  // namespace fpv
