// fpv_internal.h — declarations shared by the kernel translation units and the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpv {

template <typename T>
struct Vec4;

constexpr uint32_t kInfoTerm = 1u, kInfoTrunc = 2u, kInfoOob = 4u, kInfoFloor = 8u;  // = FPV_INFO_*

template <typename T>
struct StepArgs {
    Vec4<T>* state;        // [tiles][9][64] granules
    const T* actions;      // [n][4]
    T* obs;                // [n][20]
    T* reward;             // [n]
    uint8_t* terminated;   // [n]
    uint8_t* truncated;    // [n]
    uint64_t* tick;        // [tiles] device agent-tick counters (all equal; one per wave)
    const T* tmpl;         // reset template: 36 state words (storage-slot order) + 20 obs words
    uint32_t* led_count;   // [tiles]
    int32_t* led_idx;
    uint32_t* led_info;
    uint32_t* led_tag;
    T* led_ret;
    T* led_obs;
    int64_t led_rows;      // ledger rows per tile
    int64_t n;
    uint64_t seed;
    uint64_t id_offset;
    T sigma;
    int32_t floor_check;
    int32_t domain_rand;  // yaw task: resample the airframe at every reset
    int32_t detector;     // hover task: 0 analytic projection (SPEC §6), 1 pixel rules of hover.py:157-222 (SPEC §6b)
    // fused VecNormalize bookkeeping (all optional, fp32 handles only; see fpv_step_fused)
    uint8_t* done;       // [n] terminated | truncated
    float* ret;          // [n] discounted-return accumulator, updated in place
    double* partials;    // [tiles][kPartialWords] per-tile sums
    float gamma;
};

constexpr int kPartialWords = 44;  // sum obs[20], sum obs^2[20], sum ret, sum ret^2, 2 pad

template <typename T>
hipError_t launch_step(const StepArgs<T>& a, bool noise, hipStream_t s);
template <typename T>
hipError_t launch_reset(void* state, const void* tmpl, const uint8_t* mask, void* obs, int64_t n, hipStream_t s);
template <typename T>
hipError_t launch_init(void* state, void* tmpl, int64_t n, int detector, hipStream_t s);
template <typename T>
hipError_t launch_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s);

// fpv_yaw.hip
template <typename T>
hipError_t launch_yaw_step(const StepArgs<T>& a, bool noise, hipStream_t s);
template <typename T>
hipError_t launch_yaw_reset(void* state, const uint64_t* tick, const uint8_t* mask, void* obs, int64_t n, uint64_t seed,
                            uint64_t id_offset, int domain_rand, int all_lanes, hipStream_t s);
template <typename T>
hipError_t launch_yaw_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s);

extern int64_t g_pair_max_tiles;
int64_t step_partial_rows(int64_t n, bool noise);
int64_t yaw_partial_rows(int64_t n);
hipError_t launch_tick_fill(uint64_t* tick, int64_t tiles, uint64_t v, hipStream_t s);

// fpv_rollout.hip
hipError_t launch_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
                      const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* adv,
                      float* returns, double* stats, double* scratch, hipStream_t s);
hipError_t launch_normalize(int64_t n, int32_t d, const float* x, float* y, double* rms, int32_t update, float clip,
                            float eps, double* scratch, hipStream_t s);
hipError_t launch_moments_merge_tiles(int64_t tiles, const double* partials, double batch_count, const double* obs_rms_in,
                                      const double* ret_rms_in, double* obs_rms, double* ret_rms, double* snapshot,
                                      double* scratch, int obs_words, hipStream_t s);
hipError_t launch_reward_finalize(int64_t T, int64_t n, const float* raw, const double* snapshots, float clip, float eps,
                                  float* out, hipStream_t s);
hipError_t launch_normalize_reward(int64_t n, const float* r, const uint8_t* done, float* ret, float* r_out,
                                   double* rms, int32_t update, float gamma, float clip, float eps, double* scratch,
                                   hipStream_t s);
// Sum of v over the two 32-lane halves of the wave (lane i and lane i ^ 32), the same bits in both halves.
// v_permlane32_swap_b32 (gfx950) stays on the VALU.  The LDS route (__shfl_xor -> ds_bpermute_b32) is avoided on
// purpose: in the policy kernels it read a v_pk_fma_f32 result two instructions after the write and, with a partner
// wave streaming MFMAs on the same SIMD, sporadically saw the stale register (fpv_policy_split.hip).
__device__ __forceinline__ float half_sum(float v) {
#ifdef FPV_HALFSUM_BPERMUTE
    return v + __shfl_xor(v, 32);
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

hipError_t launch_advantage_stats(const double* gathered, int world, int stride, double* out, hipStream_t s);
hipError_t launch_zero_words(uint32_t* a, int64_t n, hipStream_t s);
hipError_t launch_ledger_scan(const uint32_t* led_count, const int32_t* led_idx, const uint32_t* led_info, const uint32_t* led_tag,
                              const float* led_ret, const float* led_obs, const uint64_t* tick, int64_t tiles, int rows,
                              int obs_words, const double* snapshots, int64_t stride, int T, float clip_obs, float eps,
                              float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                              hipStream_t s);
hipError_t launch_ledger_scatter(const int32_t* counters, const int32_t* t_idx, const int32_t* d_idx, const float* v, float gamma,
                                 float* rewards, int64_t n, int64_t cap, hipStream_t s);

// ---- VecNormalize moment merge folded into the policy kernel's prologue ----
// The rollout step is policy -> env step -> merge of the step's tile sums into the running statistics, and the next
// policy launch needs the merged statistics.  As a kernel of its own the merge costs its 7.7 us plus a launch gap on
// the critical path; here every policy workgroup folds the (<= 1024-row) table itself — same fixed order everywhere,
// so all workgroups derive the same bits — and workgroup 0 publishes the new statistics to a SECOND buffer (the old
// one is still being read by the other workgroups).
struct MergeArgs {
    const double* rows;      // [nrows][kPartialWords] tile sums of the previous env step; null = no merge
    const double* ret_rms;   // [3] {mean, var, count} before the merge, or null
    double* obs_rms_out;     // [41]; the statistics before the merge are the kernel's rms argument (may be null)
    double* ret_rms_out;     // [3] or null
    double* snapshot;        // [kPartialWords] or null
    double batch_count;
    int32_t nrows, obs_words;   // rows of 2 O + 4 words (O = obs_words <= 20)
};
constexpr int kPrologueMergeWays = 9;                                   // 9 x 44 = 396 threads of a 512-thread workgroup
constexpr int kMergeLdsBytes = kPrologueMergeWays * kPartialWords * 8;   // 3 168 B of exchange space

// Two phases so that the table's first batch travels together with whatever the caller requests in between (the
// weight image): merge_request() only issues loads, merge_prologue() adds them up and derives the statistics.
constexpr int kMergeBatch = 57;                 // rows per way and batch: one batch covers 513 rows
struct MergeRegs {
    double v[kMergeBatch];
};

__device__ __forceinline__ void merge_request(const MergeArgs& m, MergeRegs& r) {
    const int P = 2 * m.obs_words + 4;
    const int tid = threadIdx.x, j = tid / P, c = tid - j * P;
#pragma unroll
    for (int q = 0; q < kMergeBatch; ++q) {
        const int b = j + q * kPrologueMergeWays;
        r.v[q] = (j < kPrologueMergeWays && b < m.nrows) ? m.rows[(int64_t)b * P + c] : 0.0;
    }
}

// mean_f / inv_f: the normaliser the kernel will use (LDS, 20 floats each).  xch: kMergeLdsBytes of LDS.
// Contains workgroup barriers: every thread of the workgroup must call it.
__device__ __forceinline__ void merge_prologue(const MergeArgs& m, const MergeRegs& r, const double* __restrict__ obs_rms,
                                               double* xch, float eps_norm, float* mean_f, float* inv_f) {
    const int O = m.obs_words, P = 2 * O + 4;
    const int tid = threadIdx.x, j = tid / P, c = tid - j * P;
    if (j < kPrologueMergeWays) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < kMergeBatch; ++q) acc += r.v[q];
        for (int q0 = kMergeBatch; q0 * kPrologueMergeWays < m.nrows; q0 += kMergeBatch) {   // tables beyond 513 rows
            double v[kMergeBatch];
#pragma unroll
            for (int q = 0; q < kMergeBatch; ++q) {
                const int b = j + (q0 + q) * kPrologueMergeWays;
                v[q] = b < m.nrows ? m.rows[(int64_t)b * P + c] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kMergeBatch; ++q) acc += v[q];
        }
        xch[j * kPartialWords + c] = acc;
    }
    __syncthreads();
    if (tid < P) {
        double t = 0.0;
        for (int w = 0; w < kPrologueMergeWays; ++w) t += xch[w * kPartialWords + tid];
        xch[tid] = t;                           // row 0 <- column sums (each thread touches its own column only)
    }
    __syncthreads();
    const bool publish = blockIdx.x == 0;
    if (tid < O) {
        if (obs_rms) {
            const double cnt = obs_rms[2 * O], mean = obs_rms[tid], var = obs_rms[O + tid];
            const double bm = xch[tid] / m.batch_count;
            double bv = xch[O + tid] / m.batch_count - bm * bm;
            if (bv < 0.0) bv = 0.0;
            const double delta = bm - mean, tot = cnt + m.batch_count;
            const double nm = mean + delta * m.batch_count / tot;
            const double nv = (var * cnt + bv * m.batch_count + delta * delta * cnt * m.batch_count / tot) / tot;
            if (publish && m.obs_rms_out) {
                m.obs_rms_out[tid] = nm; m.obs_rms_out[O + tid] = nv;
                if (tid == 0) m.obs_rms_out[2 * O] = tot;
            }
            if (publish && m.snapshot) {
                m.snapshot[tid] = nm; m.snapshot[O + tid] = nv;
                if (tid == 0) m.snapshot[2 * O] = tot;
            }
            mean_f[tid] = (float)nm;
            inv_f[tid] = (float)(1.0 / sqrt(nv + (double)eps_norm));
        } else {
            mean_f[tid] = 0.0f;
            inv_f[tid] = 1.0f;
        }
    } else if (tid == 2 * O && m.ret_rms && publish) {
        const double cnt = m.ret_rms[2], mean = m.ret_rms[0], var = m.ret_rms[1];
        const double bm = xch[2 * O] / m.batch_count;
        double bv = xch[2 * O + 1] / m.batch_count - bm * bm;
        if (bv < 0.0) bv = 0.0;
        const double delta = bm - mean, tot = cnt + m.batch_count;
        const double nm = mean + delta * m.batch_count / tot;
        const double nv = (var * cnt + bv * m.batch_count + delta * delta * cnt * m.batch_count / tot) / tot;
        if (m.ret_rms_out) { m.ret_rms_out[0] = nm; m.ret_rms_out[1] = nv; m.ret_rms_out[2] = tot; }
        if (m.snapshot) { m.snapshot[41] = nm; m.snapshot[42] = nv; m.snapshot[43] = tot; }
    }
}

// fpv_policy.hip
hipError_t launch_policy_pack(const float* const* w, float* image, int in_dim, int act_dim, hipStream_t s);
hipError_t launch_policy_forward(const float* image, const float* obs, const double* rms, const float* eps_in,
                                 const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                 float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                 float clip_obs, float eps_norm, int deterministic, int max_blocks, int nets, int narrow,
                                 int in_dim, int act_dim, const MergeArgs& merge, const int32_t* n_dev, hipStream_t s);
int policy_image_floats();
// fpv_policy_w16.hip
hipError_t launch_policy16_pack(const float* const* w, float* image, int in_dim, int act_dim, hipStream_t s);
hipError_t launch_policy16_forward(const float* image, const float* obs, const double* rms, const uint64_t* tick, float* obs_n,
                                   float* action, float* clipped, float* value, float* logp, float* eps_out, int64_t n,
                                   uint64_t seed, uint64_t id_offset, float clip_obs, float eps_norm, int max_blocks, int in_dim,
                                   int act_dim, hipStream_t s);
int policy16_image_floats();
// fpv_ppo.hip
int64_t ppo_workspace_floats(int64_t rows, int cus);
hipError_t launch_ppo_grad(const float* const* w, float* const* grads, const int64_t* index, const float* obs,
                           const float* actions, const float* old_logp, const float* adv, const float* ret, int64_t rows,
                           float clip_range, float vf_coef, float ent_coef, float* stats, float* ws, int in_dim, int act_dim,
                           int cus, int storage, hipStream_t s);
hipError_t launch_adam_clip(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                            float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                            hipStream_t s);
// fpv_policy_split.hip
hipError_t launch_policy_split_pack(const float* const* w, unsigned char* image, hipStream_t s);
hipError_t launch_policy_split_forward(const unsigned char* image, const float* obs, const double* rms, const float* eps_in,
                                       const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                       float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                       float clip_obs, float eps_norm, int deterministic, int max_blocks,
                                       const MergeArgs& merge, hipStream_t s);
int policy_split_image_bytes();

constexpr int kScratchDoubles = 1024 * 80;  // per-device reduction scratch (partials)

}  // namespace fpv
