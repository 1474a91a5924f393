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
hipError_t launch_init(void* state, void* tmpl, int64_t n, hipStream_t s);
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
hipError_t launch_tick_fill(uint64_t* tick, int64_t tiles, uint64_t v, hipStream_t s);

// fpv_rollout.hip
hipError_t launch_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
                      const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* adv,
                      float* returns, double* stats, double* scratch, hipStream_t s);
hipError_t launch_normalize(int64_t n, int32_t d, const float* x, float* y, double* rms, int32_t update, float clip,
                            float eps, double* scratch, hipStream_t s);
hipError_t launch_moments_merge_tiles(int64_t tiles, const double* partials, double batch_count, double* obs_rms,
                                      double* ret_rms, double* snapshot, double* scratch, hipStream_t s);
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

// fpv_policy.hip
hipError_t launch_policy_pack(const float* const* w, float* image, int in_dim, int act_dim, hipStream_t s);
hipError_t launch_policy_forward(const float* image, const float* obs, const double* rms, const float* eps_in,
                                 const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                 float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                 float clip_obs, float eps_norm, int deterministic, int max_blocks, int nets, int narrow,
                                 int in_dim, int act_dim, hipStream_t s);
int policy_image_floats();
// fpv_ppo.hip
int64_t ppo_workspace_floats(int64_t rows, int cus);
hipError_t launch_ppo_grad(const float* const* w, float* const* grads, const int64_t* index, const float* obs,
                           const float* actions, const float* old_logp, const float* adv, const float* ret, int64_t rows,
                           float clip_range, float vf_coef, float ent_coef, float* stats, float* ws, int in_dim, int act_dim,
                           int cus, hipStream_t s);
hipError_t launch_adam_clip(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                            float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                            hipStream_t s);
// fpv_policy_split.hip
hipError_t launch_policy_split_pack(const float* const* w, unsigned char* image, hipStream_t s);
hipError_t launch_policy_split_forward(const unsigned char* image, const float* obs, const double* rms, const float* eps_in,
                                       const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                       float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                       float clip_obs, float eps_norm, int deterministic, int max_blocks, hipStream_t s);
int policy_split_image_bytes();

constexpr int kScratchDoubles = 1024 * 80;  // per-device reduction scratch (partials)

}  // namespace fpv
