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
    // optional: the workgroup that finishes LAST folds this launch's partial rows into the running statistics
    // (step_merge_tail below; four-tiles-per-workgroup variants only — launch_step / launch_yaw_step say whether they did)
    uint32_t* mg_ticket;  // device counter, zero between launches; null = no merge in this launch
    double* mg_obs_rms;   // [2 O + 1] {mean, var, count}, updated in place; may be null
    double* mg_ret_rms;   // [3], updated in place; may be null
    double* mg_snapshot;  // [kPartialWords] {obs statistics from word 0, ret statistics at 41..43} after the merge; may be null
    double mg_count;      // batch count of this step (= n)
};

constexpr int kPartialWords = 44;  // sum obs[20], sum obs^2[20], sum ret, sum ret^2, 2 pad

// ---- VecNormalize statistics merge by the last workgroup of the env-step launch ----
// The rollout step is policy -> env step -> RunningMeanStd.update with the step's moment sums, and the next policy
// launch needs the result.  Folding the rows in every policy workgroup's prologue (merge_prologue below) cost that
// launch 6.5 k cycles back to back and more inside the rollout graph, where the table was just written by other XCDs.
// Here the env-step workgroups publish their row write-through (sc1 stores, drained), take a ticket, and the workgroup
// whose ticket is the last one folds the whole table — fixed order, so the result does not depend on which workgroup
// it is — and updates the statistics in place; the policy launch then reads 41 doubles.  Hand-off per
// cdna_hip_programming.md Guideline 16 (R1 with the ticket form): every byte of the rows stored sc1 and drained before
// the storing workgroup's ticket add; every load of them in the reducer an sc1 load, after its own ticket add returned
// and a workgroup barrier.  The rows are rewritten by every launch; the ticket is reset by the reducer.
__device__ __forceinline__ void row_store_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double row_load_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

// Called by EVERY thread of the workgroup at the end of the kernel, after the workgroup's row has been stored with
// row_store_sc1 and each storing wave has executed s_waitcnt vmcnt(0).  xch: >= 4 P doubles of LDS.
// nrows = gridDim.x (one row per workgroup), O = observation words, P = 2 O + 4.
template <typename Args>
__device__ __forceinline__ void step_merge_tail(const Args& p, const double* rows, int O, double* xch) {
    const int P = 2 * O + 4, tid = threadIdx.x, nrows = gridDim.x;
    int* flag = reinterpret_cast<int*>(xch);
    __syncthreads();                                   // the storing wave's drain precedes the ticket (and xch is free)
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.mg_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = t == (unsigned)nrows - 1u;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();                                   // everyone has read the flag before xch is reused
    if (!last) return;
    int ways = (int)blockDim.x / P;
    if (ways > 4) ways = 4;                            // xch holds four rows
    const int j = tid / P, c = tid - j * P;
    if (j < ways) {
        double acc = 0.0;
        for (int b0 = j; b0 < nrows; b0 += 8 * ways) {  // eight rows in flight per thread, added in row order
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int b = b0 + q * ways;
                v[q] = b < nrows ? row_load_sc1(rows + (int64_t)b * P + c) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        xch[j * P + c] = acc;
    }
    __syncthreads();
    if (tid < P) {
        double t = 0.0;
        for (int w = 0; w < ways; ++w) t += xch[w * P + tid];
        xch[tid] = t;                                  // way 0's slots <- column sums (each thread its own column)
    }
    __syncthreads();
    // RunningMeanStd.update (SB3): every thread works out its new values from the old statistics first, a barrier, then
    // the stores (the count word is read by all O threads)
    const double bc = p.mg_count;
    const bool do_obs = tid < O && p.mg_obs_rms != nullptr, do_ret = tid == 2 * O && p.mg_ret_rms != nullptr;
    double nm = 0.0, nv = 0.0, tot = 0.0;
    if (do_obs || do_ret) {
        double cnt, mean, var;                          // (branches, not selects of two loads: either pointer may be null)
        if (do_obs) { cnt = p.mg_obs_rms[2 * O]; mean = p.mg_obs_rms[tid]; var = p.mg_obs_rms[O + tid]; }
        else { cnt = p.mg_ret_rms[2]; mean = p.mg_ret_rms[0]; var = p.mg_ret_rms[1]; }
        const double bm = (do_obs ? xch[tid] : xch[2 * O]) / bc;
        double bv = (do_obs ? xch[O + tid] : xch[2 * O + 1]) / bc - bm * bm;
        if (bv < 0.0) bv = 0.0;
        const double delta = bm - mean;
        tot = cnt + bc;
        nm = mean + delta * bc / tot;
        nv = (var * cnt + bv * bc + delta * delta * cnt * bc / tot) / tot;
    }
    __syncthreads();
    if (do_obs) {
        p.mg_obs_rms[tid] = nm; p.mg_obs_rms[O + tid] = nv;
        if (p.mg_snapshot) { p.mg_snapshot[tid] = nm; p.mg_snapshot[O + tid] = nv; }
        if (tid == 0) { p.mg_obs_rms[2 * O] = tot; if (p.mg_snapshot) p.mg_snapshot[2 * O] = tot; }
    }
    if (do_ret) {
        p.mg_ret_rms[0] = nm; p.mg_ret_rms[1] = nv; p.mg_ret_rms[2] = tot;
        if (p.mg_snapshot) { p.mg_snapshot[41] = nm; p.mg_snapshot[42] = nv; p.mg_snapshot[43] = tot; }
    }
    if (tid == 0) __hip_atomic_store(p.mg_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
}

// *merged (may be null) is set to true when the launch folded the statistics itself (a.mg_ticket given and honoured)
template <typename T>
hipError_t launch_step(const StepArgs<T>& a, bool noise, hipStream_t s, bool* merged = nullptr);
template <typename T>
hipError_t launch_reset(void* state, const void* tmpl, const uint8_t* mask, void* obs, int64_t n, hipStream_t s);
template <typename T>
hipError_t launch_init(void* state, void* tmpl, int64_t n, int detector, hipStream_t s);
template <typename T>
hipError_t launch_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s);

// fpv_yaw.hip
template <typename T>
hipError_t launch_yaw_step(const StepArgs<T>& a, bool noise, hipStream_t s, bool* merged = nullptr);
template <typename T>
hipError_t launch_yaw_reset(void* state, const uint64_t* tick, const uint8_t* mask, void* obs, int64_t n, uint64_t seed,
                            uint64_t id_offset, int domain_rand, int all_lanes, hipStream_t s);
template <typename T>
hipError_t launch_yaw_convert(void* state, void* canon, int64_t n, bool to_canon, hipStream_t s);

extern int64_t g_pair_max_tiles;
extern int64_t g_quad_min_tiles, g_quad_max_tiles;   // launch_step: step_quad_kernel (fpv_rollstep.hip) for tile counts in (min, max]
hipError_t launch_step_quad(const StepArgs<float>& st, uint64_t* stamps, hipStream_t s);
int64_t step_partial_rows(int64_t n, bool noise, bool pixel);
int64_t yaw_partial_rows(int64_t n);
hipError_t launch_tick_fill(uint64_t* tick, int64_t tiles, uint64_t v, hipStream_t s);

// fpv_rollout.hip
hipError_t launch_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
                      const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* adv,
                      float* returns, double* stats, double* mean_std, double* scratch, hipStream_t s);
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
hipError_t launch_rank_merge(const double* gathered, int world, int stride, int d, const double* base_obs, const double* base_ret,
                             double* out_obs, double* out_ret, hipStream_t s);
hipError_t launch_zero_words(uint32_t* a, int64_t n, hipStream_t s);
hipError_t launch_rollout_begin(uint32_t* counts, int64_t tiles, uint8_t* dst, const uint8_t* src, int64_t n, int32_t* scan_counters,
                                double* scan_sums, hipStream_t s);
hipError_t launch_ledger_pack(const uint32_t* led_count, const int32_t* led_idx, const uint32_t* led_info, const uint32_t* led_tag,
                              const float* led_ret, const float* led_obs, int64_t tiles, int rows, int obs_words, const double* rms,
                              float clip, float eps, uint32_t* out, int64_t max_rows, hipStream_t s);
hipError_t launch_ledger_scan(const uint32_t* led_count, const int32_t* led_idx, const uint32_t* led_info, const uint32_t* led_tag,
                              const float* led_ret, const float* led_obs, const uint64_t* tick, int64_t tiles, int rows,
                              int obs_words, const double* snapshots, int64_t stride, int T, float clip_obs, float eps,
                              float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                              bool counters_are_zero, uint32_t* ticket, void* status_out, const float* fin_raw, float* fin_out,
                              const double* fin_snap, int64_t fin_stride, int64_t fin_n, float fin_clip, float fin_eps, hipStream_t s);
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
struct ScatterArgs {         // fpv_policy_io.scatter_*: out[t_idx[k] * stride + d_idx[k]] += scale * value[k]; out == null: none
    const int32_t* t_idx;
    const int32_t* d_idx;
    float* out;
    int64_t stride;
    float scale;
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
                                 int in_dim, int act_dim, const MergeArgs& merge, const int32_t* n_dev, const ScatterArgs& sc,
                                 hipStream_t s);
int policy_image_floats();
// fpv_policy_w16.hip
hipError_t launch_policy16_pack(const float* const* w, float* image, int in_dim, int act_dim, hipStream_t s);
hipError_t launch_policy16_forward(const float* image, const float* obs, const double* rms, const uint64_t* tick, float* obs_n,
                                   float* action, float* clipped, float* value, float* logp, float* eps_out, int64_t n,
                                   uint64_t seed, uint64_t id_offset, float clip_obs, float eps_norm, int max_blocks, int in_dim,
                                   int act_dim, hipStream_t s);
int policy16_image_floats();
// fpv_rollstep.hip
hipError_t launch_rollout_step(const float* image, const double* rms, float* obs_n, float* action, float* clipped, float* value,
                               float* logp, uint64_t pol_seed, uint64_t pol_id_offset, float clip_obs, float eps_norm,
                               const MergeArgs& merge, const StepArgs<float>& st, int yaw, uint64_t* stamps, hipStream_t s);
hipError_t launch_rollout_step_split(const unsigned char* image, const double* rms, float* obs_n, float* action, float* clipped, float* value,
                                     float* logp, uint64_t pol_seed, uint64_t pol_id_offset, float clip_obs, float eps_norm,
                                     const MergeArgs& merge, const StepArgs<float>& st, int yaw, int with_v, uint64_t* stamps, hipStream_t s);
struct RollPersistHost {     // host-side description of a T-step launch (fpv_rollout_steps); fields as RollPersistArgs, fpv_rollstep.hip
    const float* image;
    uint64_t pol_seed, pol_id_offset;
    float clip_obs, eps_norm;
    int32_t T;
    float *obs_n, *action, *value, *logp, *reward;
    uint8_t* done;
    const double *rms0, *ret_rms0;
    double *rms_out[2], *ret_rms_out[2];
    double* snapshots;
    int64_t snap_stride;
    double* rows[2];
    uint32_t *sync, *status;
    uint64_t timeout_ticks;
    uint64_t* stamps;        // diagnostic phase stamps (fpv_debug_rollout_stamps), or null
};
hipError_t launch_rollout_persistent(const RollPersistHost& h, const StepArgs<float>& st, hipStream_t s);
int rollout_sync_words();
// fpv_ppo.hip
int64_t ppo_workspace_floats(int64_t rows, int cus);
hipError_t launch_ppo_grad(const float* const* w, float* const* grads, const int64_t* index, const float* obs,
                           const float* actions, const float* old_logp, const float* adv, const float* ret, int64_t rows,
                           float clip_range, float vf_coef, float ent_coef, float* stats, float* ws, int in_dim, int act_dim,
                           int cus, int storage, int path, hipStream_t s);
hipError_t launch_shuffle_indices(int64_t* out, int64_t n, uint64_t seed, uint64_t counter, hipStream_t s);
hipError_t launch_adam_clip(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                            float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_norm,
                            const int32_t* gate, hipStream_t s);
hipError_t launch_ppo_kl_gate(const float* stats, int n_minibatches, float limit, int32_t* gate, hipStream_t s);
// fpv_policy_split.hip
hipError_t launch_policy_split_pack(const float* const* w, unsigned char* image, int in_dim, int act_dim, hipStream_t s);
hipError_t launch_policy_split_forward(const unsigned char* image, const float* obs, const double* rms, const float* eps_in,
                                       const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                       float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                       float clip_obs, float eps_norm, int deterministic, int max_blocks, int nets, int yaw_widths,
                                       const MergeArgs& merge, hipStream_t s);
int policy_split_image_bytes();

constexpr int kScratchDoubles = 1024 * 80;  // per-device reduction scratch (partials)

}  // namespace fpv
